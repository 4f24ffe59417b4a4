// librarylink_shim.cpp -- Wolfram LibraryLink entry points over the plain C ABI (include/gphip.h).
//
// Build (on a machine with Mathematica 14+ and ROCm), against the REAL header every Wolfram installation ships
// under SystemFiles/IncludeFiles/C:
//   g++ -O2 -fPIC -shared -I$WOLFRAM/SystemFiles/IncludeFiles/C -Iinclude \
//       bayesianinference_amd/csrc/librarylink_shim.cpp -Lbayesianinference_amd/lib -lgphip \
//       -Wl,-rpath,'$ORIGIN' -o libgphip_wl.so
// In the build containers (no Wolfram installation) the same source is compiled against tests/wl_stub/
// WolframLibrary.h -- a tests-only restatement of the documented MTensor_* / MArgument_* surface over a plain
// struct -- and every entry point is driven through a fake WolframLibraryData (tests/test_gpu_wl_shim.py).
//
// Conventions honoured: "Constant" tensors are read-only and kernel-owned; results are created
// with MTensor_new and handed over with MArgument_setMTensor; strings are disowned after use; "K is not positive
// definite" is reported through the RESULT ({value, info} / info / NaN rows), never through the return code, so
// the WL closure stays numeric for every theta (BayesianStatistics.wl:276-298).  Status map: GPHIP_ERR_ARG ->
// LIBRARY_TYPE_ERROR, GPHIP_ERR_DIM -> LIBRARY_DIMENSION_ERROR, everything else -> LIBRARY_FUNCTION_ERROR; wrong
// argument count -> LIBRARY_FUNCTION_ERROR; wrong tensor rank -> LIBRARY_RANK_ERROR.
#include <algorithm>
#include <limits>
#include <cmath>
#include <cstring>
#include <vector>

#include "WolframLibrary.h"
#include "gphip.h"

static std::vector<gphip_handle> g_handles;
static std::vector<mint> g_n;              // training-set size per handle
static std::vector<mint> g_d;              // input dimension per handle (test points must have this many columns)

EXTERN_C DLLEXPORT mint WolframLibrary_getVersion() { return WolframLibraryVersion; }
// The log-prior of a JOINT (non-separable) prior reaches the native sampler as a LibraryLink callback: the reference's own
// "LogPriorPDFFunction" is a CompiledFunction of one real vector (Compile[{{param, _Real, 1}}, ..], BS:412-427), which
// ConnectLibraryCallbackFunction["gphip_logprior", f] hands to this manager; gphip_wl_nested_sampling_cb evaluates it through
// callLibraryCallbackFunction once per proposed point.
static mint g_prior_cb = 0;                       // id of the connected function (0: none)
static mbool logprior_manager(WolframLibraryData lib, mint id, MTensor argtypes) {
    if (g_prior_cb) { lib->releaseLibraryCallbackFunction(g_prior_cb); g_prior_cb = 0; }
    // exactly one argument {Real, rank 1} and a {Real, rank 0} result: rows {type, rank} of a 2 x 2 integer table
    if (lib->MTensor_getRank(argtypes) != 2 || lib->MTensor_getDimensions(argtypes)[0] != 2 || lib->MTensor_getDimensions(argtypes)[1] != 2)
        return False;
    const mint* tr = lib->MTensor_getIntegerData(argtypes);
    if (tr[0] != MType_Real || tr[1] != 1 || tr[2] != MType_Real || tr[3] != 0) return False;
    g_prior_cb = id;
    return True;
}
EXTERN_C DLLEXPORT int WolframLibrary_initialize(WolframLibraryData lib) {
    g_prior_cb = 0;
    return lib ? lib->registerLibraryCallbackManager("gphip_logprior", logprior_manager) : LIBRARY_NO_ERROR;
}
EXTERN_C DLLEXPORT void WolframLibrary_uninitialize(WolframLibraryData lib) {
    if (lib) {
        if (g_prior_cb) { lib->releaseLibraryCallbackFunction(g_prior_cb); g_prior_cb = 0; }
        lib->unregisterLibraryCallbackManager("gphip_logprior");
    }
    for (auto h : g_handles) gphip_destroy(h);
    g_handles.clear();
    g_n.clear();
    g_d.clear();
}

// a real tensor of the given rank; 0 = fine, else the LIBRARY_* code to return
static int want_real(WolframLibraryData lib, MTensor t, mint rank) {
    if (lib->MTensor_getRank(t) != rank) return LIBRARY_RANK_ERROR;
    if (lib->MTensor_getType(t) != MType_Real) return LIBRARY_TYPE_ERROR;
    return 0;
}
// optional per-point array: an EMPTY list stands for "the constant form" (null pointer at the C ABI); otherwise a real
// tensor with exactly `count` elements
static int optional_values(WolframLibraryData lib, MTensor t, mint count, const double** out) {
    *out = nullptr;
    if (lib->MTensor_getFlattenedLength(t) == 0) return 0;
    if (lib->MTensor_getType(t) != MType_Real) return LIBRARY_TYPE_ERROR;
    if (lib->MTensor_getFlattenedLength(t) != count) return LIBRARY_DIMENSION_ERROR;
    *out = lib->MTensor_getRealData(t);
    return 0;
}

static gphip_handle lookup(mint id) {
    return (id >= 0 && (size_t)id < g_handles.size()) ? g_handles[(size_t)id] : nullptr;
}

static int status_to_wl(int rc) {
    switch (rc) {
        case GPHIP_OK: return LIBRARY_NO_ERROR;
        case GPHIP_ERR_ARG: return LIBRARY_TYPE_ERROR;
        case GPHIP_ERR_DIM: return LIBRARY_DIMENSION_ERROR;
        default: return LIBRARY_FUNCTION_ERROR;
    }
}

// gphip_wl_create[X (N x d), y (N), kernelId, meanId, dtype (64 | 32), devices (integer list, may be empty)]
// -> handle id.  Several device ordinals = one multi-device handle (gphip_create, include/gphip.h).
EXTERN_C DLLEXPORT int gphip_wl_create(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 6) return LIBRARY_FUNCTION_ERROR;
    MTensor X = MArgument_getMTensor(args[0]), y = MArgument_getMTensor(args[1]), dv = MArgument_getMTensor(args[5]);
    if (lib->MTensor_getRank(X) != 2 || lib->MTensor_getRank(y) != 1 || lib->MTensor_getRank(dv) != 1) return LIBRARY_RANK_ERROR;
    if (lib->MTensor_getType(X) != MType_Real || lib->MTensor_getType(y) != MType_Real ||
        lib->MTensor_getType(dv) != MType_Integer)
        return LIBRARY_TYPE_ERROR;
    const mint* dims = lib->MTensor_getDimensions(X);
    if (lib->MTensor_getDimensions(y)[0] != dims[0]) return LIBRARY_DIMENSION_ERROR;
    const mint ndev = lib->MTensor_getDimensions(dv)[0];
    std::vector<int> devs((size_t)ndev);
    for (mint i = 0; i < ndev; ++i) devs[(size_t)i] = (int)lib->MTensor_getIntegerData(dv)[i];
    gphip_handle h = nullptr;
    int rc = gphip_create(lib->MTensor_getRealData(X), lib->MTensor_getRealData(y), dims[0], dims[1],
                          (int)MArgument_getInteger(args[2]), (int)MArgument_getInteger(args[3]),
                          (int)MArgument_getInteger(args[4]), ndev ? devs.data() : nullptr, (int)ndev, &h);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    g_handles.push_back(h);
    g_n.push_back(dims[0]);
    g_d.push_back(dims[1]);
    MArgument_setInteger(res, (mint)g_handles.size() - 1);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_kernel_spec[name, d] -> {kernelId, term1, op, term2, offset, params of term 1, of term 2, 0-based index of sigma_n}
// (gphip_kernel_parse: the kernel-name grammar lives in the library, not in the package); {-1, ..} = not a kernel of the library
EXTERN_C DLLEXPORT int gphip_wl_kernel_spec(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    char* name = MArgument_getUTF8String(args[0]);
    int spec[8];
    const int rc = gphip_kernel_parse(name, (int64_t)MArgument_getInteger(args[1]), spec);
    lib->UTF8String_disown(name);
    MTensor r; mint dims[1] = {8};
    if (lib->MTensor_new(MType_Integer, 1, dims, &r)) return LIBRARY_FUNCTION_ERROR;
    for (int i = 0; i < 8; ++i) lib->MTensor_getIntegerData(r)[i] = rc == GPHIP_OK ? spec[i] : -1;
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_create_custom[X (N x d), y (N), cform, nparams, meanId, dtype, devices (integer list, may be empty)] -> handle id.
// ANY `kernel @@ points[[{i,j}]]` of the reference (BGP:29-33): GPHIP.wl applies the pure function to two points of stand-in
// symbols (..gphipXc<k>, ..gphipYc<k>; hyper-parameters ..gphipPc<k>) and hands over ToString[CForm[..]] AS IT IS; the
// translation to the function body (gphip_cform_to_body) and the run-time compilation into the library's kernel build happen
// here.  Text that is no C expression, or does not compile, returns LIBRARY_FUNCTION_ERROR (the package then falls back to the
// reference's own path; gphip_create_error() has the compiler's log).
EXTERN_C DLLEXPORT int gphip_wl_create_custom(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 7) return LIBRARY_FUNCTION_ERROR;
    MTensor X = MArgument_getMTensor(args[0]), y = MArgument_getMTensor(args[1]), dv = MArgument_getMTensor(args[6]);
    if (lib->MTensor_getRank(X) != 2 || lib->MTensor_getRank(y) != 1 || lib->MTensor_getRank(dv) != 1) return LIBRARY_RANK_ERROR;
    if (lib->MTensor_getType(X) != MType_Real || lib->MTensor_getType(y) != MType_Real || lib->MTensor_getType(dv) != MType_Integer)
        return LIBRARY_TYPE_ERROR;
    const mint* dims = lib->MTensor_getDimensions(X);
    if (lib->MTensor_getDimensions(y)[0] != dims[0]) return LIBRARY_DIMENSION_ERROR;
    const mint ndev = lib->MTensor_getDimensions(dv)[0];
    std::vector<int> devs((size_t)ndev);
    for (mint i = 0; i < ndev; ++i) devs[(size_t)i] = (int)lib->MTensor_getIntegerData(dv)[i];
    char* cform = MArgument_getUTF8String(args[2]);
    std::vector<char> body(strlen(cform) + 64);
    int rc = gphip_cform_to_body(cform, body.data(), (int64_t)body.size());
    lib->UTF8String_disown(cform);
    if (rc != GPHIP_OK) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = nullptr;
    rc = gphip_create_custom_devices(lib->MTensor_getRealData(X), lib->MTensor_getRealData(y), dims[0], dims[1], body.data(),
                                     (int)MArgument_getInteger(args[3]), (int)MArgument_getInteger(args[4]),
                                     (int)MArgument_getInteger(args[5]), ndev ? devs.data() : nullptr, (int)ndev, &h);
    if (rc == GPHIP_ERR_ARG) return LIBRARY_FUNCTION_ERROR;          // the body does not compile (gphip_create_error() has the log)
    if (rc != GPHIP_OK) return status_to_wl(rc);
    g_handles.push_back(h);
    g_n.push_back(dims[0]);
    g_d.push_back(dims[1]);
    MArgument_setInteger(res, (mint)g_handles.size() - 1);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_set_option[h, name, value] -> 0
EXTERN_C DLLEXPORT int gphip_wl_set_option(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 3) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    char* name = MArgument_getUTF8String(args[1]);
    int rc = h ? gphip_set_option(h, name, MArgument_getReal(args[2])) : GPHIP_ERR_STATE;
    lib->UTF8String_disown(name);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MArgument_setInteger(res, 0);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_loglik[h, theta] -> {value, info}
EXTERN_C DLLEXPORT int gphip_wl_loglik(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (lib->MTensor_getRank(th) != 1) return LIBRARY_RANK_ERROR;
    double out = 0.0; int info = 0;
    int rc = gphip_loglik(h, lib->MTensor_getRealData(th), (int)lib->MTensor_getDimensions(th)[0], &out, &info);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MTensor r; mint d[1] = {2};
    if (lib->MTensor_new(MType_Real, 1, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* p = lib->MTensor_getRealData(r);
    p[0] = info == 0 ? out : 0.0; p[1] = (double)info;
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_loglik_batch[h, Theta (B x p)] -> B x 2 {{value, info}..}
EXTERN_C DLLEXPORT int gphip_wl_loglik_batch(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (lib->MTensor_getRank(th) != 2) return LIBRARY_RANK_ERROR;
    const mint* dims = lib->MTensor_getDimensions(th);
    std::vector<double> out((size_t)dims[0]);
    std::vector<int> info((size_t)dims[0]);
    int rc = gphip_loglik_batch(h, lib->MTensor_getRealData(th), (int)dims[0], (int)dims[1], out.data(), info.data());
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MTensor r; mint d[2] = {dims[0], 2};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* p = lib->MTensor_getRealData(r);
    for (mint i = 0; i < dims[0]; ++i) { p[2 * i] = info[(size_t)i] == 0 ? out[(size_t)i] : 0.0; p[2 * i + 1] = info[(size_t)i]; }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_loglik_grad[h, theta] -> {value, info, d/dtheta_1 .. d/dtheta_p}   (gradient entries NaN when info != 0)
EXTERN_C DLLEXPORT int gphip_wl_loglik_grad(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (lib->MTensor_getRank(th) != 1) return LIBRARY_RANK_ERROR;
    const mint p = lib->MTensor_getDimensions(th)[0];
    MTensor r; mint d[1] = {p + 2};
    if (lib->MTensor_new(MType_Real, 1, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* o = lib->MTensor_getRealData(r);
    double out = 0.0; int info = 0;
    int rc = gphip_loglik_grad(h, lib->MTensor_getRealData(th), (int)p, &out, o + 2, &info);
    if (rc != GPHIP_OK) { lib->MTensor_free(r); return status_to_wl(rc); }
    o[0] = info == 0 ? out : 0.0; o[1] = (double)info;
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_fit[h, theta] -> info
EXTERN_C DLLEXPORT int gphip_wl_fit(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (lib->MTensor_getRank(th) != 1) return LIBRARY_RANK_ERROR;
    int info = 0;
    int rc = gphip_fit(h, lib->MTensor_getRealData(th), (int)lib->MTensor_getDimensions(th)[0], &info);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MArgument_setInteger(res, info);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_solve[h, b]: b a vector (N) or a matrix (N x m) -> K^-1 b, same shape ("Inverse" of
// matrixInverseAndDet accepts both, BayesianGaussianProcess.wl:194, 410, 416).  Needs a successful fit.
EXTERN_C DLLEXPORT int gphip_wl_solve(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor b = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    const mint rank = lib->MTensor_getRank(b);
    if (rank != 1 && rank != 2) return LIBRARY_RANK_ERROR;
    const mint* dims = lib->MTensor_getDimensions(b);
    const mint N = g_n[(size_t)id], m = rank == 2 ? dims[1] : 1;
    if (dims[0] != N) return LIBRARY_DIMENSION_ERROR;
    const double* src = lib->MTensor_getRealData(b);
    // the C ABI wants each right-hand side contiguous (column-major N x m); a WL matrix is row-major N x m
    std::vector<double> in((size_t)N * m), out((size_t)N * m);
    for (mint i = 0; i < N; ++i)
        for (mint j = 0; j < m; ++j) in[(size_t)(j * N + i)] = src[i * m + j];
    int rc = gphip_solve(h, in.data(), m, out.data());
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MTensor r;
    if (lib->MTensor_new(MType_Real, rank, dims, &r)) return LIBRARY_FUNCTION_ERROR;
    double* o = lib->MTensor_getRealData(r);
    for (mint i = 0; i < N; ++i)
        for (mint j = 0; j < m; ++j) o[i * m + j] = out[(size_t)(j * N + i)];
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_logdet[h] -> log det K of the fitted theta
EXTERN_C DLLEXPORT int gphip_wl_logdet(WolframLibraryData, mint argc, MArgument* args, MArgument res) {
    if (argc != 1) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    if (!h) return LIBRARY_FUNCTION_ERROR;
    double v = 0.0;
    int rc = gphip_logdet(h, &v);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MArgument_setReal(res, v);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_predict[h, Xs (M x d)] -> 2 x M {means, variances}
EXTERN_C DLLEXPORT int gphip_wl_predict(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor xs = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, xs, 2)) return e;
    if (lib->MTensor_getDimensions(xs)[1] != g_d[(size_t)id]) return LIBRARY_DIMENSION_ERROR;   // the C ABI reads M * d doubles
    const mint M = lib->MTensor_getDimensions(xs)[0];
    MTensor r; mint d[2] = {2, M};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* p = lib->MTensor_getRealData(r);
    int rc = gphip_predict(h, lib->MTensor_getRealData(xs), M, p, p + M);
    if (rc != GPHIP_OK) { lib->MTensor_free(r); return status_to_wl(rc); }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_predict_samples[h, Thetas (S x p), Xs (M x d)] -> 2 x S x M {means, variances};
// a sample whose K is not positive definite comes back as NaN rows (result value, not an error)
EXTERN_C DLLEXPORT int gphip_wl_predict_samples(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 3) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor th = MArgument_getMTensor(args[1]), xs = MArgument_getMTensor(args[2]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, th, 2)) return e;
    if (int e = want_real(lib, xs, 2)) return e;
    if (lib->MTensor_getDimensions(xs)[1] != g_d[(size_t)id]) return LIBRARY_DIMENSION_ERROR;
    const mint S = lib->MTensor_getDimensions(th)[0], p = lib->MTensor_getDimensions(th)[1];
    const mint M = lib->MTensor_getDimensions(xs)[0];
    MTensor r; mint d[3] = {2, S, M};
    if (lib->MTensor_new(MType_Real, 3, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* out = lib->MTensor_getRealData(r);
    std::vector<int> info((size_t)S);
    int rc = gphip_predict_samples(h, lib->MTensor_getRealData(th), (int)S, (int)p, lib->MTensor_getRealData(xs), M,
                                   out, out + S * M, info.data());
    if (rc != GPHIP_OK) { lib->MTensor_free(r); return status_to_wl(rc); }
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (mint s = 0; s < S; ++s)
        if (info[(size_t)s] != 0)
            for (mint t = 0; t < M; ++t) out[s * M + t] = out[S * M + s * M + t] = nan;
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_covariance[h, theta] -> N x N
EXTERN_C DLLEXPORT int gphip_wl_covariance(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 2) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (lib->MTensor_getRank(th) != 1) return LIBRARY_RANK_ERROR;
    const mint N = g_n[(size_t)id];
    MTensor r; mint d[2] = {N, N};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    int rc = gphip_covariance(h, lib->MTensor_getRealData(th), (int)lib->MTensor_getDimensions(th)[0],
                              lib->MTensor_getRealData(r));
    if (rc != GPHIP_OK) { lib->MTensor_free(r); return status_to_wl(rc); }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_cross_covariance[h, theta, Xs (M x d)] -> (N + 1) x M: rows 0..N-1 = k (compiledKandKappa's "k",
// BayesianGaussianProcess.wl:100-109), last row = kappa (:110-115)
EXTERN_C DLLEXPORT int gphip_wl_cross_covariance(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 3) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor th = MArgument_getMTensor(args[1]), xs = MArgument_getMTensor(args[2]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, th, 1)) return e;
    if (int e = want_real(lib, xs, 2)) return e;
    if (lib->MTensor_getDimensions(xs)[1] != g_d[(size_t)id]) return LIBRARY_DIMENSION_ERROR;
    const mint N = g_n[(size_t)id], M = lib->MTensor_getDimensions(xs)[0];
    MTensor r; mint d[2] = {N + 1, M};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* o = lib->MTensor_getRealData(r);
    int rc = gphip_cross_covariance(h, lib->MTensor_getRealData(th), (int)lib->MTensor_getDimensions(th)[0],
                                    lib->MTensor_getRealData(xs), M, o, o + N * M);
    if (rc != GPHIP_OK) { lib->MTensor_free(r); return status_to_wl(rc); }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_device_count[] -> number of gfx950 devices visible to this process (sub-kernels of parallelNestedSampling
// pick Mod[$KernelID, count], BayesianStatistics.wl:1349-1357)
EXTERN_C DLLEXPORT int gphip_wl_device_count(WolframLibraryData, mint argc, MArgument*, MArgument res) {
    if (argc != 0) return LIBRARY_FUNCTION_ERROR;
    int n = 0;
    gphip_device_count(&n);
    MArgument_setInteger(res, n);
    return LIBRARY_NO_ERROR;
}

// ---- point-dependent nugget[x] / meanFunction[x] (BayesianGaussianProcess.wl:37, 113, 171, 300, 408): WL maps the two
// functions over the points for the theta of the call and passes the VALUES; an empty list = the constant form.
// gphip_wl_loglik_batch_pw[h, Theta (B x p), meanTrain (B x N | {}), nuggetTrain (B x N | {})] -> B x 2 {{value, info}..}
EXTERN_C DLLEXPORT int gphip_wl_loglik_batch_pw(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 4) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, th, 2)) return e;
    const mint* dims = lib->MTensor_getDimensions(th);
    const double *mt = nullptr, *nt = nullptr;
    if (int e = optional_values(lib, MArgument_getMTensor(args[2]), dims[0] * g_n[(size_t)id], &mt)) return e;
    if (int e = optional_values(lib, MArgument_getMTensor(args[3]), dims[0] * g_n[(size_t)id], &nt)) return e;
    std::vector<double> out((size_t)dims[0]);
    std::vector<int> info((size_t)dims[0]);
    int rc = gphip_loglik_batch_pw(h, lib->MTensor_getRealData(th), (int)dims[0], (int)dims[1], mt, nt, out.data(), info.data());
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MTensor r; mint d[2] = {dims[0], 2};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* p = lib->MTensor_getRealData(r);
    for (mint i = 0; i < dims[0]; ++i) { p[2 * i] = info[(size_t)i] == 0 ? out[(size_t)i] : 0.0; p[2 * i + 1] = info[(size_t)i]; }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_fit_pw[h, theta, meanTrain (N | {}), nuggetTrain (N | {})] -> info
EXTERN_C DLLEXPORT int gphip_wl_fit_pw(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 4) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor th = MArgument_getMTensor(args[1]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, th, 1)) return e;
    const double *mt = nullptr, *nt = nullptr;
    if (int e = optional_values(lib, MArgument_getMTensor(args[2]), g_n[(size_t)id], &mt)) return e;
    if (int e = optional_values(lib, MArgument_getMTensor(args[3]), g_n[(size_t)id], &nt)) return e;
    int info = 0;
    int rc = gphip_fit_pw(h, lib->MTensor_getRealData(th), (int)lib->MTensor_getDimensions(th)[0], mt, nt, &info);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MArgument_setInteger(res, info);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_predict_samples_pw[h, Thetas (S x p), meanTrain (S x N | {}), nuggetTrain (S x N | {}), Xs (M x d),
//                             meanTest (S x M | {}), nuggetTest (S x M | {})] -> 2 x S x M {means, variances}
EXTERN_C DLLEXPORT int gphip_wl_predict_samples_pw(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 7) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    MTensor th = MArgument_getMTensor(args[1]), xs = MArgument_getMTensor(args[4]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, th, 2)) return e;
    if (int e = want_real(lib, xs, 2)) return e;
    if (lib->MTensor_getDimensions(xs)[1] != g_d[(size_t)id]) return LIBRARY_DIMENSION_ERROR;
    const mint S = lib->MTensor_getDimensions(th)[0], p = lib->MTensor_getDimensions(th)[1];
    const mint M = lib->MTensor_getDimensions(xs)[0], N = g_n[(size_t)id];
    const double *mt = nullptr, *nt = nullptr, *ms = nullptr, *nsx = nullptr;
    if (int e = optional_values(lib, MArgument_getMTensor(args[2]), S * N, &mt)) return e;
    if (int e = optional_values(lib, MArgument_getMTensor(args[3]), S * N, &nt)) return e;
    if (int e = optional_values(lib, MArgument_getMTensor(args[5]), S * M, &ms)) return e;
    if (int e = optional_values(lib, MArgument_getMTensor(args[6]), S * M, &nsx)) return e;
    MTensor r; mint d[3] = {2, S, M};
    if (lib->MTensor_new(MType_Real, 3, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* out = lib->MTensor_getRealData(r);
    std::vector<int> info((size_t)S);
    int rc = gphip_predict_samples_pw(h, lib->MTensor_getRealData(th), (int)S, (int)p, mt, nt, lib->MTensor_getRealData(xs), M, ms,
                                      nsx, out, out + S * M, info.data());
    if (rc != GPHIP_OK) { lib->MTensor_free(r); return status_to_wl(rc); }
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (mint s = 0; s < S; ++s)
        if (info[(size_t)s] != 0)
            for (mint t = 0; t < M; ++t) out[s * M + t] = out[S * M + s * M + t] = nan;
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_nested_sampling[h, box (p x 2), priorKinds (p integers: 0 uniform, 1 log-uniform), opts (real vector:
//   pool, maxIterations, minIterations, mcSteps, walkers, terminationFraction, minAccept, maxAccept, seed),
//   start (pool x p | {})] -> n x (p + 3): {point.., logLikelihood, logPriorPDF, acceptanceRate (NaN for the pool)} per
// sample in generation order.  The native driver of nestedSamplingInternal (BayesianStatistics.wl:859-1040); the WL
// package wraps the rows into the reference's "Samples" association and calls the reference's evidenceSampling on it.
EXTERN_C DLLEXPORT int gphip_wl_nested_sampling(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 5) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor box = MArgument_getMTensor(args[1]), kinds = MArgument_getMTensor(args[2]), ov = MArgument_getMTensor(args[3]),
            st = MArgument_getMTensor(args[4]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, box, 2)) return e;
    if (int e = want_real(lib, ov, 1)) return e;
    if (lib->MTensor_getRank(kinds) != 1) return LIBRARY_RANK_ERROR;
    if (lib->MTensor_getType(kinds) != MType_Integer) return LIBRARY_TYPE_ERROR;
    int p = 0;
    gphip_num_params(h, &p);
    if (lib->MTensor_getDimensions(box)[0] != p || lib->MTensor_getDimensions(box)[1] != 2 || lib->MTensor_getDimensions(kinds)[0] != p ||
        lib->MTensor_getDimensions(ov)[0] != 9)
        return LIBRARY_DIMENSION_ERROR;
    const double* o = lib->MTensor_getRealData(ov);
    gphip_ns_options opt;
    gphip_ns_default_options(&opt);
    opt.pool = (int)o[0]; opt.max_iterations = (int)o[1]; opt.min_iterations = (int)o[2]; opt.mc_steps = (int)o[3];
    opt.walkers = (int)o[4]; opt.termination_fraction = o[5]; opt.min_accept = o[6]; opt.max_accept = o[7]; opt.seed = (uint64_t)o[8];
    if (opt.pool < 2) return LIBRARY_DIMENSION_ERROR;
    const double* start = nullptr;
    if (int e = optional_values(lib, st, (mint)opt.pool * p, &start)) return e;
    std::vector<int> kd((size_t)p);
    for (int j = 0; j < p; ++j) kd[(size_t)j] = (int)lib->MTensor_getIntegerData(kinds)[j];
    const int64_t cap = (int64_t)opt.pool + std::max(opt.max_iterations, opt.min_iterations) + 1;
    std::vector<double> pts((size_t)cap * p), ll((size_t)cap), lp((size_t)cap), ar((size_t)cap);
    int64_t n = 0, ne = 0;
    double z = 0.0;
    int rc = gphip_nested_sampling(h, lib->MTensor_getRealData(box), kd.data(), nullptr, nullptr, &opt, start, cap, pts.data(), ll.data(),
                                   lp.data(), ar.data(), &n, &z, &ne);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MTensor r; mint d[2] = {(mint)n, (mint)p + 3};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* out = lib->MTensor_getRealData(r);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < p; ++j) out[i * (p + 3) + j] = pts[(size_t)(i * p + j)];
        out[i * (p + 3) + p] = ll[(size_t)i];
        out[i * (p + 3) + p + 1] = lp[(size_t)i];
        out[i * (p + 3) + p + 2] = ar[(size_t)i];
    }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// ---- any SEPARABLE prior for the native sampler (BayesianStatistics.wl:256-274: "LogPriorPDFFunction" of a
// ProductDistribution of univariate distributions -- Normal, LogNormal, Gamma, .. -- is the sum of the factors' log densities).
// The WL package tabulates each factor's log density on a uniform grid over the parameter's {min, max} (paramSpecPattern,
// BS:19) and draws the starting pool itself (RandomVariate[prior, pool], BS:1046-1068); the library's C ABI takes the prior
// as its gphip_logprior_fn callback -- implemented HERE by interpolating the tables (4-point Lagrange, error O(h^4)).
namespace {
struct TabPrior { int p; mint m; const double* box; const double* tab; };
double tab_logprior(const double* th, int p, void* user) {
    const TabPrior* t = static_cast<const TabPrior*>(user);
    double s = 0.0;
    for (int j = 0; j < p; ++j) {
        const double lo = t->box[2 * j], hi = t->box[2 * j + 1];
        const double u = (th[j] - lo) / (hi - lo) * (double)(t->m - 1);
        mint i = (mint)std::floor(u);
        if (i < 1) i = 1;
        if (i > t->m - 3) i = t->m - 3;
        const double x = u - (double)i;                       // nodes at -1, 0, 1, 2 relative to i
        const double* f = t->tab + (size_t)j * t->m + (i - 1);
        auto zero = [](double v) { return !(v > -1e290); };      // -1e300 (or NaN / -inf) marks a zero of the density
        if (zero(f[0]) || zero(f[1]) || zero(f[2]) || zero(f[3])) {
            // next to a zero of the density (log = -inf): no polynomial through it -- the nearer node decides
            const double v = f[1 + (x > 0.5 ? 1 : 0)];
            if (zero(v)) return -INFINITY;
            s += v;
            continue;
        }
        s += -x * (x - 1.0) * (x - 2.0) / 6.0 * f[0] + (x + 1.0) * (x - 1.0) * (x - 2.0) / 2.0 * f[1] -
             (x + 1.0) * x * (x - 2.0) / 2.0 * f[2] + (x + 1.0) * x * (x - 1.0) / 6.0 * f[3];
    }
    return s;
}
}  // namespace

// gphip_wl_nested_sampling_tab[h, box (p x 2), logPriorTables (p x m, m >= 4: log density of factor j at the m grid nodes of
//   [min_j, max_j]), opts (as gphip_wl_nested_sampling), start (pool x p, or {} = the library draws opts[[1]] points from the
//   tabulated prior itself: gphip_tab_prior_sample, seeded with opts[[9]])]  -> n x (p + 3) rows as gphip_wl_nested_sampling
EXTERN_C DLLEXPORT int gphip_wl_nested_sampling_tab(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 5) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor box = MArgument_getMTensor(args[1]), tab = MArgument_getMTensor(args[2]), ov = MArgument_getMTensor(args[3]),
            st = MArgument_getMTensor(args[4]);
    if (!h) return LIBRARY_FUNCTION_ERROR;
    if (int e = want_real(lib, box, 2)) return e;
    if (int e = want_real(lib, tab, 2)) return e;
    if (int e = want_real(lib, ov, 1)) return e;
    const bool draw = lib->MTensor_getRank(st) == 1 && lib->MTensor_getDimensions(st)[0] == 0;      // {}: no starting points given
    if (!draw)
        if (int e = want_real(lib, st, 2)) return e;
    int p = 0;
    gphip_num_params(h, &p);
    const mint m = lib->MTensor_getDimensions(tab)[1];
    if (lib->MTensor_getDimensions(box)[0] != p || lib->MTensor_getDimensions(box)[1] != 2 || lib->MTensor_getDimensions(tab)[0] != p ||
        m < 4 || lib->MTensor_getDimensions(ov)[0] != 9 || (!draw && lib->MTensor_getDimensions(st)[1] != p))
        return LIBRARY_DIMENSION_ERROR;
    const double* o = lib->MTensor_getRealData(ov);
    gphip_ns_options opt;
    gphip_ns_default_options(&opt);
    opt.pool = draw ? (int)o[0] : (int)lib->MTensor_getDimensions(st)[0]; opt.max_iterations = (int)o[1]; opt.min_iterations = (int)o[2];
    opt.mc_steps = (int)o[3];
    opt.walkers = (int)o[4]; opt.termination_fraction = o[5]; opt.min_accept = o[6]; opt.max_accept = o[7]; opt.seed = (uint64_t)o[8];
    if (opt.pool < 2) return LIBRARY_DIMENSION_ERROR;
    TabPrior tp{p, m, lib->MTensor_getRealData(box), lib->MTensor_getRealData(tab)};
    std::vector<double> drawn;
    if (draw) {                                            // generateStartingPoints (BS:1046-1068): the pool comes from the prior
        drawn.resize((size_t)opt.pool * p);
        if (gphip_tab_prior_sample(tp.box, tp.tab, p, (int64_t)m, opt.pool, opt.seed, drawn.data()) != GPHIP_OK) return LIBRARY_NUMERICAL_ERROR;
    }
    const int64_t cap = (int64_t)opt.pool + std::max(opt.max_iterations, opt.min_iterations) + 1;
    std::vector<double> pts((size_t)cap * p), ll((size_t)cap), lp((size_t)cap), ar((size_t)cap);
    int64_t n = 0, ne = 0;
    double z = 0.0;
    int rc = gphip_nested_sampling(h, tp.box, nullptr, tab_logprior, &tp, &opt, draw ? drawn.data() : lib->MTensor_getRealData(st), cap, pts.data(), ll.data(),
                                   lp.data(), ar.data(), &n, &z, &ne);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    MTensor r; mint d[2] = {(mint)n, (mint)p + 3};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* out = lib->MTensor_getRealData(r);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < p; ++j) out[i * (p + 3) + j] = pts[(size_t)(i * p + j)];
        out[i * (p + 3) + p] = ll[(size_t)i];
        out[i * (p + 3) + p + 1] = lp[(size_t)i];
        out[i * (p + 3) + p + 2] = ar[(size_t)i];
    }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

// gphip_wl_nested_sampling_cb[h, box (p x 2), opts (as gphip_wl_nested_sampling), start (pool x p: drawn from the prior by the caller,
//   generateStartingPoints BS:1046-1068)] -> n x (p + 3) rows as gphip_wl_nested_sampling.  The prior is the function connected with
//   ConnectLibraryCallbackFunction["gphip_logprior", logPriorPDFFunction] -- ANY "LogPriorPDFFunction" (BS:256-274), joint priors
//   included; values <= -1e300 (the reference's $MachineLogZero outside the constraints), NaN or a failing evaluation read as log 0.
namespace {
struct CbPrior { WolframLibraryData lib; MTensor arg; int failed; };
double cb_logprior(const double* th, int p, void* user) {
    CbPrior* c = static_cast<CbPrior*>(user);
    double* dst = c->lib->MTensor_getRealData(c->arg);
    for (int j = 0; j < p; ++j) dst[j] = th[j];
    mreal value = 0.0;
    MArgument args[1], res;
    MArgument_getMTensorAddress(args[0]) = &c->arg;
    MArgument_getRealAddress(res) = &value;
    if (c->lib->callLibraryCallbackFunction(g_prior_cb, 1, args, res) != LIBRARY_NO_ERROR) { c->failed = 1; return -INFINITY; }
    return (value > -1e300) ? value : -INFINITY;             // (NaN compares false: log 0)
}
}  // namespace
EXTERN_C DLLEXPORT int gphip_wl_nested_sampling_cb(WolframLibraryData lib, mint argc, MArgument* args, MArgument res) {
    if (argc != 4) return LIBRARY_FUNCTION_ERROR;
    gphip_handle h = lookup(MArgument_getInteger(args[0]));
    MTensor box = MArgument_getMTensor(args[1]), ov = MArgument_getMTensor(args[2]), st = MArgument_getMTensor(args[3]);
    if (!h || !g_prior_cb) return LIBRARY_FUNCTION_ERROR;   // (no function connected: ConnectLibraryCallbackFunction first)
    if (int e = want_real(lib, box, 2)) return e;
    if (int e = want_real(lib, ov, 1)) return e;
    if (int e = want_real(lib, st, 2)) return e;
    int p = 0;
    gphip_num_params(h, &p);
    if (lib->MTensor_getDimensions(box)[0] != p || lib->MTensor_getDimensions(box)[1] != 2 || lib->MTensor_getDimensions(ov)[0] != 9 ||
        lib->MTensor_getDimensions(st)[1] != p)
        return LIBRARY_DIMENSION_ERROR;
    const double* o = lib->MTensor_getRealData(ov);
    gphip_ns_options opt;
    gphip_ns_default_options(&opt);
    opt.pool = (int)lib->MTensor_getDimensions(st)[0]; opt.max_iterations = (int)o[1]; opt.min_iterations = (int)o[2]; opt.mc_steps = (int)o[3];
    opt.walkers = (int)o[4]; opt.termination_fraction = o[5]; opt.min_accept = o[6]; opt.max_accept = o[7]; opt.seed = (uint64_t)o[8];
    if (opt.pool < 2) return LIBRARY_DIMENSION_ERROR;
    CbPrior cb{lib, nullptr, 0};
    mint pd[1] = {(mint)p};
    if (lib->MTensor_new(MType_Real, 1, pd, &cb.arg)) return LIBRARY_FUNCTION_ERROR;
    const int64_t cap = (int64_t)opt.pool + std::max(opt.max_iterations, opt.min_iterations) + 1;
    std::vector<double> pts((size_t)cap * p), ll((size_t)cap), lp((size_t)cap), ar((size_t)cap);
    int64_t n = 0, ne = 0;
    double z = 0.0;
    const int rc = gphip_nested_sampling(h, lib->MTensor_getRealData(box), nullptr, cb_logprior, &cb, &opt, lib->MTensor_getRealData(st), cap,
                                         pts.data(), ll.data(), lp.data(), ar.data(), &n, &z, &ne);
    lib->MTensor_free(cb.arg);
    if (rc != GPHIP_OK) return status_to_wl(rc);
    if (cb.failed) return LIBRARY_FUNCTION_ERROR;           // the connected function could not be evaluated
    MTensor r; mint d[2] = {(mint)n, (mint)p + 3};
    if (lib->MTensor_new(MType_Real, 2, d, &r)) return LIBRARY_FUNCTION_ERROR;
    double* out = lib->MTensor_getRealData(r);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < p; ++j) out[i * (p + 3) + j] = pts[(size_t)(i * p + j)];
        out[i * (p + 3) + p] = ll[(size_t)i];
        out[i * (p + 3) + p + 1] = lp[(size_t)i];
        out[i * (p + 3) + p + 2] = ar[(size_t)i];
    }
    MArgument_setMTensor(res, r);
    return LIBRARY_NO_ERROR;
}

EXTERN_C DLLEXPORT int gphip_wl_destroy(WolframLibraryData, mint argc, MArgument* args, MArgument res) {
    if (argc != 1) return LIBRARY_FUNCTION_ERROR;
    const mint id = MArgument_getInteger(args[0]);
    gphip_handle h = lookup(id);
    if (h) { gphip_destroy(h); g_handles[(size_t)id] = nullptr; }
    MArgument_setInteger(res, 0);
    return LIBRARY_NO_ERROR;
}
