/* WolframLibrary.h -- TESTS-ONLY STAND-IN.  NOT the Wolfram header and not ABI compatible with it.
 *
 * The real WolframLibrary.h ships with every Wolfram installation (SystemFiles/IncludeFiles/C) and is absent from
 * the build containers.  This file restates, over a plain struct, just the DOCUMENTED LibraryLink surface that
 * bayesianinference_amd/csrc/librarylink_shim.cpp uses (names, argument meaning and return conventions as in the
 * LibraryLink user guide: "Library Structure and Life Cycle", "Interaction with Wolfram Language", MTensor /
 * MArgument reference pages), so that the shim can be COMPILED and every gphip_wl_* entry point DRIVEN through a
 * fake WolframLibraryData (tests/wl_stub/shim_driver.cpp, tests/test_gpu_wl_shim.py).  A production build uses the
 * real header; nothing under bayesianinference_amd/ includes this file. */
#ifndef GPHIP_TEST_WOLFRAMLIBRARY_H
#define GPHIP_TEST_WOLFRAMLIBRARY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define EXTERN_C extern "C"
#else
#define EXTERN_C
#endif
#define DLLEXPORT __attribute__((visibility("default")))

#define WolframLibraryVersion 7

typedef int64_t mint;
typedef double mreal;
typedef int mbool;
typedef struct { mreal ri[2]; } mcomplex;

/* library function return codes */
#define LIBRARY_NO_ERROR 0
#define LIBRARY_TYPE_ERROR 1
#define LIBRARY_RANK_ERROR 2
#define LIBRARY_DIMENSION_ERROR 3
#define LIBRARY_NUMERICAL_ERROR 4
#define LIBRARY_MEMORY_ERROR 5
#define LIBRARY_FUNCTION_ERROR 6
#define LIBRARY_VERSION_ERROR 7

/* MTensor element types */
#define MType_Integer 2
#define MType_Real 3
#define MType_Complex 4

typedef struct st_MTensor {          /* opaque in the real header */
    mint type, rank, flat_length;
    mint dims[8];
    void* data;
    int constant;                    /* passed "Constant": the library must not free or write it */
} * MTensor;

typedef union {
    mbool* boolean;
    mint* integer;
    mreal* real;
    mcomplex* cmplex;
    MTensor* tensor;
    char** utf8string;
} MArgument;

#define MArgument_getInteger(a) (*((a).integer))
#define MArgument_getReal(a) (*((a).real))
#define MArgument_getMTensor(a) (*((a).tensor))
#define MArgument_getUTF8String(a) (*((a).utf8string))
#define MArgument_setInteger(a, v) ((*((a).integer)) = (v))
#define MArgument_setReal(a, v) ((*((a).real)) = (v))
#define MArgument_setMTensor(a, v) ((*((a).tensor)) = (v))
#define MArgument_getRealAddress(a) ((a).real)
#define MArgument_getMTensorAddress(a) ((a).tensor)
#define True 1
#define False 0

struct st_WolframLibraryData;
typedef struct st_WolframLibraryData* WolframLibraryData;
struct st_WolframLibraryData {
    void (*UTF8String_disown)(char*);
    int (*MTensor_new)(mint type, mint rank, const mint* dims, MTensor* out);
    void (*MTensor_free)(MTensor);
    mint (*MTensor_getRank)(MTensor);
    const mint* (*MTensor_getDimensions)(MTensor);
    mint (*MTensor_getType)(MTensor);
    mint (*MTensor_getFlattenedLength)(MTensor);
    mint* (*MTensor_getIntegerData)(MTensor);
    mreal* (*MTensor_getRealData)(MTensor);
    void (*Message)(const char*);
    mint (*AbortQ)(void);
    /* "Callback Evaluations" (LibraryLink user guide): ConnectLibraryCallbackFunction["name", compiledFunction] makes the kernel
     * call the manager registered under "name" with an id and the (nargs + 1) x 2 {type, rank} table of the function's arguments
     * and result; the library then evaluates the function with callLibraryCallbackFunction(id, ..) */
    int (*registerLibraryCallbackManager)(const char* name, mbool (*mfun)(WolframLibraryData, mint, MTensor));
    int (*unregisterLibraryCallbackManager)(const char* name);
    int (*callLibraryCallbackFunction)(mint id, mint ArgC, MArgument* Args, MArgument Res);
    int (*releaseLibraryCallbackFunction)(mint id);
};

#endif
