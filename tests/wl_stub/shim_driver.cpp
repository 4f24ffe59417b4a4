// shim_driver.cpp -- TESTS-ONLY fake WolframLibraryData behind tests/wl_stub/WolframLibrary.h.
// Linked into the same shared object as the shim (built by bayesianinference_amd.build.build_wl_stub); the pytest
// side (tests/test_gpu_wl_shim.py) creates tensors, builds MArgument arrays with ctypes and calls the gphip_wl_*
// entry points exactly as the Wolfram kernel would.  It counts live library-created tensors and disowned strings so
// the tests can check the ownership conventions (results handed over, temporaries freed on error paths).
#include <cstdlib>
#include <cstring>

#include "WolframLibrary.h"

static long g_live = 0, g_disowned = 0, g_const_frees = 0;

static int t_new(mint type, mint rank, const mint* dims, MTensor* out) {
    if (rank < 0 || rank > 8 || (type != MType_Integer && type != MType_Real)) return LIBRARY_TYPE_ERROR;
    MTensor t = static_cast<MTensor>(calloc(1, sizeof(*t)));
    t->type = type; t->rank = rank; t->flat_length = 1;
    for (mint i = 0; i < rank; ++i) { t->dims[i] = dims[i]; t->flat_length *= dims[i]; }
    t->data = calloc((size_t)(t->flat_length > 0 ? t->flat_length : 1), 8);
    ++g_live;
    *out = t;
    return LIBRARY_NO_ERROR;
}
static void t_free(MTensor t) {
    if (!t) return;
    if (t->constant) { ++g_const_frees; return; }      // a library must never free a "Constant" argument
    free(t->data); free(t);
    --g_live;
}
static mint t_rank(MTensor t) { return t->rank; }
static const mint* t_dims(MTensor t) { return t->dims; }
static mint t_type(MTensor t) { return t->type; }
static mint t_flat(MTensor t) { return t->flat_length; }
static mint* t_idata(MTensor t) { return static_cast<mint*>(t->data); }
static mreal* t_rdata(MTensor t) { return static_cast<mreal*>(t->data); }
static void s_disown(char*) { ++g_disowned; }
static void msg(const char*) {}
static mint abortq(void) { return 0; }

// ---- callback evaluations: one manager per name, one connected function (a C function pointer standing in for the
// CompiledFunction: theta (Real, rank 1) -> Real) per id
typedef mbool (*manager_fn)(WolframLibraryData, mint, MTensor);
typedef double (*compiled_fn)(const double* theta, mint n);
static struct { char name[64]; manager_fn fn; } g_managers[8];
static int g_nmanagers = 0;
static compiled_fn g_connected[16];
static mint g_next_id = 1;
static long g_released = 0, g_cb_calls = 0;
static int cb_register(const char* name, manager_fn fn) {
    for (int i = 0; i < g_nmanagers; ++i)               // (a library initialised twice in one process registers the same name again)
        if (!strcmp(g_managers[i].name, name)) { g_managers[i].fn = fn; return LIBRARY_NO_ERROR; }
    if (g_nmanagers >= 8) return LIBRARY_FUNCTION_ERROR;
    strncpy(g_managers[g_nmanagers].name, name, 63);
    g_managers[g_nmanagers++].fn = fn;
    return LIBRARY_NO_ERROR;
}
static int cb_unregister(const char* name) {
    for (int i = 0; i < g_nmanagers; ++i)
        if (!strcmp(g_managers[i].name, name)) { g_managers[i] = g_managers[--g_nmanagers]; return LIBRARY_NO_ERROR; }
    return LIBRARY_FUNCTION_ERROR;
}
static int cb_call(mint id, mint argc, MArgument* args, MArgument res) {
    if (id < 1 || id >= 16 || !g_connected[id] || argc != 1) return LIBRARY_FUNCTION_ERROR;
    MTensor t = MArgument_getMTensor(args[0]);
    if (!t || t->type != MType_Real || t->rank != 1) return LIBRARY_TYPE_ERROR;
    ++g_cb_calls;
    *MArgument_getRealAddress(res) = g_connected[id](static_cast<const double*>(t->data), t->dims[0]);
    return LIBRARY_NO_ERROR;
}
static int cb_release(mint id) {
    if (id >= 1 && id < 16 && g_connected[id]) { g_connected[id] = nullptr; ++g_released; }
    return LIBRARY_NO_ERROR;
}

static st_WolframLibraryData g_data = {s_disown, t_new, t_free, t_rank, t_dims, t_type, t_flat, t_idata, t_rdata, msg, abortq,
                                       cb_register, cb_unregister, cb_call, cb_release};

extern "C" {
DLLEXPORT WolframLibraryData drv_libdata(void) { return &g_data; }
// an argument tensor as the kernel would pass it with "Constant": data copied in, flagged read-only
DLLEXPORT MTensor drv_tensor(mint type, mint rank, const mint* dims, const void* data) {
    MTensor t = nullptr;
    if (t_new(type, rank, dims, &t)) return nullptr;
    --g_live;                                           // kernel-owned, not a library allocation
    if (data) memcpy(t->data, data, (size_t)t->flat_length * 8);
    t->constant = 1;
    return t;
}
DLLEXPORT void drv_release(MTensor t) {                 // the kernel releasing an argument / a received result
    if (!t) return;
    if (!t->constant) --g_live;
    free(t->data); free(t);
}
DLLEXPORT mint drv_rank(MTensor t) { return t->rank; }
DLLEXPORT mint drv_type(MTensor t) { return t->type; }
DLLEXPORT const mint* drv_dims(MTensor t) { return t->dims; }
DLLEXPORT void* drv_data(MTensor t) { return t->data; }
DLLEXPORT long drv_live(void) { return g_live; }
DLLEXPORT long drv_disowned(void) { return g_disowned; }
DLLEXPORT long drv_const_frees(void) { return g_const_frees; }
// ConnectLibraryCallbackFunction[name, f] as the kernel does it: the manager gets a fresh id and the {type, rank} table
// (here: one Real vector argument, a Real scalar result -- or, wrong = 1, a table the manager must refuse); returns the
// manager's verdict (1 = connected), -1 = no such manager
DLLEXPORT int drv_connect_callback(const char* name, compiled_fn f, int wrong) {
    for (int i = 0; i < g_nmanagers; ++i)
        if (!strcmp(g_managers[i].name, name)) {
            const mint id = g_next_id < 16 ? g_next_id++ : 15;
            g_connected[id] = f;
            const mint dims[2] = {2, 2};
            const mint table[4] = {MType_Real, wrong ? 2 : 1, MType_Real, 0};
            MTensor t = drv_tensor(MType_Integer, 2, dims, table);
            const mbool ok = g_managers[i].fn(&g_data, id, t);
            drv_release(t);
            if (!ok) g_connected[id] = nullptr;
            return ok ? 1 : 0;
        }
    return -1;
}
DLLEXPORT long drv_cb_released(void) { return g_released; }
DLLEXPORT long drv_cb_calls(void) { return g_cb_calls; }
}
