/* gphip.h -- plain C ABI of the MI355X-native Gaussian-process likelihood / prediction path.
 *
 * Drop-in boundary for the GP hot path of ssmit1986/BayesianInference.  Every
 * entry point replaces a Wolfram-Language call site of the reference
 * (BayesianInference/Kernel/BayesianGaussianProcess.wl, cited as BGP:line):
 *
 *   gphip_create        data capture of defineGaussianProcess              BGP:228-238
 *   gphip_loglik        the "LogLikelihoodFunction" closure theta -> R     BGP:295-306 (+181-199)
 *   gphip_loglik_batch  the same closure mapped over a theta matrix        BS:276-298, BS:902-916
 *   gphip_fit           matrixInverseAndDet[covarianceFunction[theta]]     BGP:308 (invCovFun)
 *   gphip_cross_covariance  compiledKandKappa[points1, kernel, nugget][X*]     BGP:63-124
 *   gphip_predict       predictFromGaussianProcessInternal                 BGP:396-422
 *   gphip_predict_samples  predictFromGaussianProcess over all samples     BGP:343-376
 *   gphip_*_pw          the same with point-dependent nugget[x] / mean[x]  BGP:37, 113, 171, 300, 408
 *   gphip_covariance    "CovarianceFunction" = compiledCovarianceMatrix    BGP:45-61
 *   gphip_solve         "InverseCovarianceFunction"[theta]["Inverse"][b]   BGP:130-141
 *   gphip_logdet        "InverseCovarianceFunction"[theta]["LogDet"]       BGP:126-128,139
 *
 * Conventions: every function returns an int status (0 = GPHIP_OK).  "K is not positive
 * definite / hopelessly ill-conditioned" is NOT an error status: it is reported through *info
 * (the WL shim / Python host then substitutes $MachineLogZero exactly like Catch["MatInv"],
 * BGP:298-304).  theta and results are fp64 at the ABI.  Host buffers stay owned by the caller;
 * X and y are copied to the device once and stay resident.  One in-flight call per handle
 * (internally serialised); different handles may be used concurrently.
 */
#ifndef GPHIP_H
#define GPHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gphip_ctx* gphip_handle;

/* status codes */
#define GPHIP_OK 0
#define GPHIP_ERR_ARG 1        /* bad argument value / unknown id            (LIBRARY_TYPE_ERROR)      */
#define GPHIP_ERR_DIM 2        /* shape mismatch (p, N, d, M)                (LIBRARY_DIMENSION_ERROR) */
#define GPHIP_ERR_HIP 3        /* HIP runtime failure, see gphip_last_error  (LIBRARY_FUNCTION_ERROR)  */
#define GPHIP_ERR_STATE 4      /* predict/solve/logdet before a successful gphip_fit                    */
#define GPHIP_ERR_NODEVICE 5   /* no gfx950 device visible                                              */
#define GPHIP_ERR_UNSUPPORTED 6

/* *info values */
#define GPHIP_INFO_OK 0
#define GPHIP_INFO_NOT_SPD 1   /* pivot <= tol: singular or ill-conditioned K (LinearSolve::sing1/luc) */
#define GPHIP_INFO_NAN 2       /* non-finite theta or result                                           */

/* kernel_id: named covariance functions (exact forms in SURVEY.md §8d / DESIGN.md).
 * theta layout: (l_1..l_nl, sigma_f, sigma_n [, mu]);  nl = 1 (isotropic) or d (ARD). */
#define GPHIP_KERNEL_SE 0            /* sf^2 exp(-|p-q|^2 / (2 l^2))                                  */
#define GPHIP_KERNEL_SE_ARD 1        /* sf^2 exp(-1/2 sum ((p_j-q_j)/l_j)^2)                          */
#define GPHIP_KERNEL_MATERN52 2      /* sf^2 (1+sqrt5 s+5 s^2/3) exp(-sqrt5 s), s=|p-q|/l             */
#define GPHIP_KERNEL_MATERN52_ARD 3  /* same with s = sqrt(sum ((p_j-q_j)/l_j)^2)                     */
#define GPHIP_KERNEL_NULL 4          /* Function[0]: K = diag(sn^2) (BGP:25-27,156-159); theta=(sn[,mu]) */
#define GPHIP_KERNEL_MATERN32 5      /* sf^2 (1 + sqrt3 s) exp(-sqrt3 s), s = |p-q|/l                  */
#define GPHIP_KERNEL_MATERN32_ARD 6
#define GPHIP_KERNEL_RQ 7            /* rational quadratic sf^2 (1 + r2/(2 alpha))^-alpha, r2 = |p-q|^2/l^2; theta = (l, alpha, sf, ..) */
#define GPHIP_KERNEL_RQ_ARD 8        /* same with r2 = sum ((p_j-q_j)/l_j)^2;                  theta = (l_1..l_d, alpha, sf, ..)  */
/* The reference takes ANY kernel[p, q] (BGP:32); its own worked example is a constant plus a squared exponential,
 * `#2 + Exp[-(pt1 - pt2)^2/#1^2]` (BGP:16).  Composed forms of the named kernels cover that family:
 *     k = [c +] k1   |   [c +] k1 + k2   |   [c +] k1 * k2
 * each term with its own length scales / alpha / sf.  theta layout:
 *     [term 1: l.., (alpha), sf] [term 2: l.., (alpha), sf] [c] sn [mu]
 * (c = the constant offset, present when offset = 1).  Anything else stays on the reference's own path (the WL package
 * falls back to the reference's defineGaussianProcess).  */
#define GPHIP_OP_NONE 0
#define GPHIP_OP_SUM 1
#define GPHIP_OP_PRODUCT 2
#define GPHIP_KERNEL_COMPOSE(term1, op, term2, offset) \
    ((term1) | ((term2) << 8) | ((op) << 16) | ((offset) << 20) | (1 << 24))
/* ANY other covariance function: source text compiled at run time (hiprtc) into the library's own kernel build.
 * `body` = the statements of
 *     template <typename T> T k(X, Y, P, D) { <body> }
 * where X(k) / Y(k) are coordinate k (0-based) of the two points, P(k) the function's k-th hyper-parameter, D the input
 * dimension and T the handle's arithmetic type (double / float); it must `return` the covariance WITHOUT the nugget.  Device
 * math (exp, sqrt, pow, fabs, sin, ..) and the names Mathematica's CForm emits (Power, Sqrt, Exp, Log, Abs, Sin, Cos, Tan, ArcTan, Sinh, Cosh, Tanh, Erf, Erfc, Min, Max, Pi, E)
 * are available.  Example, SE-ARD:  "T s = 0; for (int k = 0; k < D; ++k) { T u = (X(k) - Y(k)) / P(k); s += u * u; }
 * return P(D) * P(D) * exp((T)-0.5 * s);"  with nparams = d + 1.
 * theta layout of such a handle:  [p_0 .. p_{nparams-1}] sn [mu].  The function may be non-stationary: the prior variance
 * k(x, x) is evaluated per point on the device (prediction variance, pivot tolerance of the factorisation).
 * Likelihoods, batches, fits, predictions, posterior-sample mixtures, covariance exports and the native sampler work as for the
 * named kernels.  gphip_loglik_grad costs ONE factorisation, like the named kernels: the body is instantiated a second time
 * with T = a forward-mode dual number (csrc/gp_dual.h: + - * / comparisons, exp log log1p expm1 sqrt pow fabs sin cos tan
 * sinh cosh tanh atan erf erfc fmin fmax and the CForm names) inside the reduction 1/2 tr((alpha alpha^T - K^-1) dK/dp_m);
 * intermediates that depend on a P(k) must therefore be of type T (not double / float).  A body that does not compile
 * that way, more than 64 hyper-parameters, or option "custom_grad" = 0 fall back to central differences of the likelihood
 * (2 p + 1 points in one batched evaluation; step eps^(1/3) max(|theta_k|, 1e-2) in the handle's arithmetic: ~1e-6 relative
 * in fp64, ~1e-2 in fp32).  Option "grad_analytic" reads 1 after a call that took the one-factorisation route.  Replaces the reference's
 * `kernel @@ points[[{i,j}]]` for an arbitrary pure function (BGP:29-33, cross form BGP:100-109).
 * Errors: GPHIP_ERR_ARG = the body does not compile (gphip_create_error() returns the compiler's log),
 * GPHIP_ERR_UNSUPPORTED = no hiprtc (see csrc/rtc_dyn.h; $GPHIP_HIPRTC_PATH names the library explicitly). */
#define GPHIP_KERNEL_CUSTOM 100
int gphip_create_custom(const void* X, const void* y, int64_t N, int64_t d, const char* body, int nparams, int mean_id,
                        int dtype, int device /* < 0: current */, gphip_handle* out);
/* the same with a device list (several ordinals = ONE multi-device handle, see gphip_create) / as one rank of a multi-process
 * job (see gphip_create_rank): every local context compiles the function for itself */
int gphip_create_custom_devices(const void* X, const void* y, int64_t N, int64_t d, const char* body, int nparams, int mean_id,
                                int dtype, const int* devices, int ndev, gphip_handle* out);
int gphip_create_custom_rank(const void* X, const void* y, int64_t N, int64_t d, const char* body, int nparams, int mean_id,
                             int dtype, int device, int rank, int world, const void* id128, gphip_handle* out);
const char* gphip_create_error(void);
/* gphip_custom_compile_d: compile `body` exactly as gphip_create_custom of a d-dimensional problem would (the program is
 * specialised on the input dimension for 1 <= d <= 32; d = 0 or d > 32: the dimension-generic program) -- no handle, no device
 * needed (hiprtc cross-compiles for `arch`, null = "gfx950").  Validates user input early; proves that a deployed libgphip.so finds hiprtc and carries its own kernel text (the
 * text of csrc/gp_kernels.h is embedded in the library at build time; $GPHIP_SRC_DIR overrides it for development); and warms
 * the per-process code-object cache (key: body, dtype, arch, d) that later gphip_create_custom* calls OF THE SAME d hit.  *cache_hit = 1 when
 * the code object was already there.  grad_nparams < 0: the value program (kernel build + prior variance kernels);
 * 1 .. 64: the GRADIENT program instead -- the same text instantiated with forward-mode dual numbers in its grad_nparams
 * hyper-parameters (csrc/gp_dual.h), what the first gphip_loglik_grad of such a handle compiles.
 * Errors as gphip_create_custom (gphip_create_error() = the compiler's log). */
int gphip_custom_compile_d(const char* body, int dtype, const char* arch, int grad_nparams, int d, int* cache_hit);
/* the same with d = 0: validates the function and warms the cache of handles with d > 32 only */
int gphip_custom_compile(const char* body, int dtype, const char* arch, int grad_nparams, int* cache_hit);

/* ---- host logic shared by the two hosts (pure C++, no device; csrc/gphip_hostlogic.inc) ------------------------------------
 * gphip_kernel_parse: the kernel-name grammar (spaces, underscores, case ignored)
 *     term [(+|*) term] [+ const]    term: se, se_ard, matern52, matern52_ard, matern32, matern32_ard, rq, rq_ard;   null | none
 * -> spec[8] = {kernel_id for gphip_create, term 1, op, term 2 (-1 none), offset, parameters of term 1, of term 2, 0-based
 * position of sigma_n in theta}.  GPHIP_ERR_ARG: not a kernel of the library.
 * gphip_cform_to_body: the text Mathematica's CForm prints for a covariance function whose coordinates / hyper-parameters
 * were replaced by the stand-in symbols ..gphipXc<k> / ..gphipYc<k> / ..gphipPc<k>  ->  "return <C expression in X(k), Y(k),
 * P(k)>;", the body gphip_create_custom takes (BGP:29-33: any pure function of two points).
 * gphip_tab_prior_sample: `pool` draws from a separable prior given as log-density tables (p x m, uniform nodes over
 * box[2j] .. box[2j+1]; <= -1e299 = log 0): the starting pool of the sampler (generateStartingPoints, BS:1046-1068). */
int gphip_kernel_parse(const char* name, int64_t d, int* spec);
int gphip_cform_to_body(const char* cform, char* out, int64_t cap);
int gphip_tab_prior_sample(const double* box, const double* tab, int p, int64_t m, int pool, uint64_t seed, double* out);
#define GPHIP_MEAN_ZERO 0            /* Function[0]  (BGP:168,255)                                    */
#define GPHIP_MEAN_CONST 1           /* Function[mu], mu = last entry of theta                        */

/* X: row-major N x d, y: length N, always fp64 host arrays.  dtype selects the DEVICE arithmetic:
 * 64 = fp64 (v_mfma_f64_16x16x4_f64, the 1e-8 parity path) or 32 = fp32 (v_mfma_f32_16x16x4_f32,
 * BASELINE.json config 5; ~1e-3 relative vs the fp64 oracle).  theta and every result stay fp64.
 * devices/ndev: HIP device ordinals this handle may use.  NULL/0 = the current device; one ordinal = that
 * device; SEVERAL ordinals = a multi-device handle living in this one process (one context per listed
 * device, communicator = RCCL ncclCommInitAll, bound at run time; a repeated ordinal makes several virtual
 * ranks share a GPU -- device copies instead of RCCL -- which is how the sharded schedule is tested on a
 * 1-GPU box).  Every entry point below works unchanged on a multi-device handle:
 *   gphip_loglik / _parts / gphip_fit   N >= option "shard_min_n" (default 16384): ONE factorisation sharded
 *                                       over the devices, 1-D block-cyclic over outer panels, one broadcast of
 *                                       the factored panel per step; smaller N: first device
 *   gphip_loglik_batch, gphip_predict_samples   thetas / posterior samples dealt to the devices, no collective
 *   gphip_predict                       after a sharded fit every device holds the whole factor (each received
 *                                       panel is unpacked on arrival): test points shard, no collective
 *   everything else                     first device
 * The reference-side caller this serves: ONE WL kernel process (nestedSampling, BS:1099-1136). */
int gphip_create(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id,
                 int dtype, const int* devices, int ndev, gphip_handle* out);
int gphip_destroy(gphip_handle h);

/* One process PER device (WL sub-kernels of parallelNestedSampling, BS:1349-1357; torchrun ranks): rank 0 calls
 * gphip_comm_unique_id, the host hands the GPHIP_COMM_ID_BYTES bytes to every process by its own means, and each
 * process creates its handle with gphip_create_rank (collective: ncclCommInitRank).  Afterwards gphip_loglik /
 * gphip_fit on these handles are COLLECTIVE calls (every rank, same theta, same order) whenever
 * N >= "shard_min_n"; results are identical on every rank.  GPHIP_ERR_UNSUPPORTED if no RCCL can be bound. */
#define GPHIP_COMM_ID_BYTES 128
int gphip_comm_unique_id(void* id128);
int gphip_create_rank(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id, int dtype,
                      int device, int rank, int world, const void* id128, gphip_handle* out);
/* world = ranks of the job (1 for a plain handle), nlocal = ranks in this process, *comm = "none" /
 * "device copies" / "rccl (..)" (string owned by the library); any pointer may be NULL */
int gphip_comm_info(gphip_handle h, int* world, int* nlocal, const char** comm);

/* number of hyper-parameters p expected for this handle */
int gphip_num_params(gphip_handle h, int* p);

/* log p(y | X, theta) = -1/2 (N log 2pi + log det K + r^T K^-1 r).  *out undefined if *info != 0. */
int gphip_loglik(gphip_handle h, const double* theta, int p, double* out, int* info);
/* Theta: row-major B x p; out, info: length B. */
int gphip_loglik_batch(gphip_handle h, const double* Theta, int B, int p, double* out, int* info);
/* loglik plus its parts: parts[0]=log det K, parts[1]=r^T K^-1 r (for parity tests). */
int gphip_loglik_parts(gphip_handle h, const double* theta, int p, double* out, double* parts,
                       int* info);

/* loglik and d loglik / d theta (grad: length p, same order as theta):
 * 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_p).  No reference counterpart (the
 * reference maximises without gradients, LaplaceApproximation.wl:177-238).  grad = NaN if *info != 0. */
int gphip_loglik_grad(gphip_handle h, const double* theta, int p, double* out, double* grad, int* info);

/* Factor K(theta) and keep L and L^-1 r resident for predict / solve / logdet. */
int gphip_fit(gphip_handle h, const double* theta, int p, int* info);
/* Xs: row-major M x d fp64.  mean[j] = m(x*_j) + k*_j^T K^-1 r ; var[j] = k(x*,x*) + sn^2 -
 * k*_j^T K^-1 k*_j  (variance of a noisy observation, BGP:113,414-417; sd = sqrt(var)). */
int gphip_predict(gphip_handle h, const void* Xs, int64_t M, double* mean, double* var);
/* predictFromGaussianProcess over posterior samples (BGP:343-376): Thetas row-major S x p; mean, var
 * row-major S x M; info[S] (a sample with info != 0 has undefined mean/var).  All samples of a chunk
 * are factored and solved in one batched pass (one workspace slot per sample). */
int gphip_predict_samples(gphip_handle h, const double* Thetas, int S, int p, const void* Xs, int64_t M,
                          double* mean, double* var, int* info);
/* ---- Point-dependent nugget and mean functions.  The reference evaluates nugget[points[[i]]] (BGP:37), meanFunction /@
 * inputData (BGP:171, 300), kernel[p, p] + nugget[p] at the test points (BGP:113) and meanFunction /@ inputs (BGP:408) for
 * ANY functions of the point (heteroscedastic noise, any m(x)).  Those functions live on the host (WL / Python); the host
 * evaluates them for the theta(s) of the call and hands the VALUES over:
 *   mean_train, nugget_train   row-major B x N (S x N for the samples form): m_theta_b(x_i), nu_theta_b(x_i) at the training
 *                              points; nugget values are VARIANCES (they take the place of sn^2 on the diagonal)
 *   mean_test,  nugget_test    length M (gphip_predict_pw) / row-major S x M (samples form): the same at the test points
 * A NULL pointer keeps the constant form read from theta (sn^2 resp. mu / 0).  theta keeps its layout; the entries a vector
 * replaces are not used (sn^2 still scales nothing: the pivot tolerance uses sf^2 + max|nugget|).  Arrays are read during
 * the call only.  Gradients (gphip_loglik_grad) are defined for the constant forms only. */
int gphip_loglik_batch_pw(gphip_handle h, const double* Theta, int B, int p, const double* mean_train,
                          const double* nugget_train, double* out, int* info);
int gphip_fit_pw(gphip_handle h, const double* theta, int p, const double* mean_train, const double* nugget_train, int* info);
int gphip_predict_pw(gphip_handle h, const void* Xs, int64_t M, const double* mean_test, const double* nugget_test,
                     double* mean, double* var);
int gphip_predict_samples_pw(gphip_handle h, const double* Thetas, int S, int p, const double* mean_train,
                             const double* nugget_train, const void* Xs, int64_t M, const double* mean_test,
                             const double* nugget_test, double* mean, double* var, int* info);
/* K: row-major N x N fp64 (full, both triangles), for parity tests at small N. */
int gphip_covariance(gphip_handle h, const double* theta, int p, double* K);
/* Listable form (BGP:59): Theta row-major B x p -> K row-major B x N x N. */
int gphip_covariance_batch(gphip_handle h, const double* Theta, int B, int p, double* K);
/* compiledKandKappa (BGP:91-124, null kernel BGP:63-89): k row-major N x M (rows = training points, columns =
 * test points, the layout of BGP:103-107), kappa[M] = k(x*,x*) + nugget.  No fit needed; un-fits the handle. */
int gphip_cross_covariance(gphip_handle h, const double* theta, int p, const void* Xs, int64_t M, double* k,
                           double* kappa);
/* rhs, out: column-major N x nrhs (each right-hand side contiguous); out = K^-1 rhs. */
int gphip_solve(gphip_handle h, const double* rhs, int64_t nrhs, double* out);
int gphip_logdet(gphip_handle h, double* out);

/* ---- Nested sampling over the hyper-parameters, driven natively.  Restates the reference's
 * nestedSamplingInternal (BayesianStatistics.wl:859-1040): same live-point bookkeeping, X values (:790-802), trapezoid
 * weights (:757-771), stop rule (:967-978) and constrained-prior Metropolis move (:707-745) -- but `walkers` chains advance in
 * lock step so that every Metropolis step is ONE batched likelihood call instead of "MonteCarloSteps" sequential ones.
 *   box          p x 2 row-major {min, max} per parameter (paramSpecPattern, BS:19)
 *   prior_kind   per parameter 0 = uniform, 1 = log-uniform over its range (NULL = all uniform); or
 *   logprior     a "LogPriorPDFFunction" callback (BS:256-274) overriding prior_kind; the box still bounds the walk
 *   start        pool x p starting points (generateStartingPoints, BS:1046-1068) or NULL = drawn from the built-in prior
 * Output, in generation order (the pool first, then one sample per iteration), up to `cap` samples: points (cap x p),
 * loglik, logprior_out, accept_rate (NaN for the pool: Missing["InitialSample"]); *n_samples; *log_evidence = the crude
 * estimate logSumExp of the crude posterior weights (BS:1019); *n_evals = likelihood evaluations spent.  The statistical
 * post-processing (evidenceSampling, BS:1158-1291; combineRuns) stays with the host: the WL package hands the samples to
 * the reference's OWN evidenceSampling. */
typedef double (*gphip_logprior_fn)(const double* theta, int p, void* user);
typedef struct {
    int pool;                      /* "SamplePoolSize", default 100 (BS:833-855) */
    int max_iterations;            /* "MaxIterations" 10000 */
    int min_iterations;            /* "MinIterations" 100 */
    int mc_steps;                  /* "MonteCarloSteps" 200 */
    int walkers;                   /* chains advanced in lock step, default 32 (1 = the reference's sequential chain) */
    double termination_fraction;   /* "TerminationFraction" 0.01 */
    double min_accept, max_accept; /* "MinMaxAcceptanceRate" {0, 1} */
    uint64_t seed;
} gphip_ns_options;
int gphip_ns_default_options(gphip_ns_options* opts);
int gphip_nested_sampling(gphip_handle h, const double* box, const int* prior_kind, gphip_logprior_fn logprior, void* user,
                          const gphip_ns_options* opts, const double* start, int64_t cap, double* points, double* loglik,
                          double* logprior_out, double* accept_rate, int64_t* n_samples, double* log_evidence,
                          int64_t* n_evals);
/* calculateWeightsCrude (BS:818-835) for m samples of which `pool` are live: order (sorted by (LogLikelihood, Point)), log X
 * and crude log posterior weights in that order, and their logSumExp (exported for tests and hosts without the reference). */
int gphip_ns_crude_weights(const double* points, const double* loglik, int64_t m, int p, int pool, int64_t* order,
                           double* logx, double* logw, double* log_evidence);

/* Options (tuning knobs; results do not depend on them beyond rounding order):
 *   "panel"        outer panel width in 128-tiles (default 4)
 *   "panel_wide"   0/1 (default 1): single-device schedule uses 2x / 1.5x that width while >= 192 / >= 128 tile columns
 *                  remain (the long trailing update hides the wider panel and is read-modify-written less often)
 *   "profile"      0 off, 1 kernel build + trailing SYRK + whole evaluation + prediction epilogue, 2 every kernel class
 *   "xcd_swizzle"  0/1 XCD-aware tile order of the GEMM launches (default 1)
 *   "supertile"    tile order of the trailing SYRK: 0 column-major list, one contiguous chunk per XCD; 1 whole 8x8 super-tiles
 *                  dealt statically to the XCDs (measured slower: unequal loads); 2 (default) the tile LIST itself in 8x8
 *                  super-tile order, equal contiguous chunks per XCD (16 operand panels per 64 resident tiles instead of 65:
 *                  less cache-to-L2 traffic, SYRK alone 0.849 -> 0.861 of peak); 3 = 2, for batches of thetas too (measured: no
 *                  difference there)
 *   "lookahead"    0/1 factor panel k+1 on a second stream under the trailing update of panel k (default 1)
 *   "latency_gemm" 0/1 4x4-wave GEMM shape for launches of <= 256 tiles of problems up to "latency_max_nt" tile columns
 *   "dataflow"     0/1 single-launch dataflow Cholesky (one workgroup per tile, flags instead of launches)
 *                  for problems of <= "dataflow_max_nt" 128-tiles (default 96, N <= 12288), with 64x64 tiles up to
 *                  "dataflow_fine_nt" 128-tiles (default 96, fp64).  How many thetas of a call share ONE such launch:
 *                  "dataflow_max_slots" -1 (default) = fp64: as many as keep the launch within "dataflow_max_tasks" 64-tile
 *                  tasks (default 34 000, the measured crossover with the multi-kernel batch at every N = 512 .. 12288: 750
 *                  thetas at N = 512, 128 at 1024, 16 at 4096, 4 at 8192), fp32 (128-tiles): 2 / 5 of that many 128-tile tasks; n >= 1 = at most n thetas (and up to 4 n
 *                  of a problem with <= 2 500 tasks in all: the rule before round 6).  Larger problems hand their last
 *                  "dataflow_tail" tile columns (default 64, 0 = off) to the same kernel
 *   "panel_left"   -1 auto / 0 / 1: left-looking in-panel updates (one K = 128 s update per column instead of
 *                  K = 128 updates after every column); auto = for batches of more than 8 thetas ("dataflow_max_slots" if set)
 *                  and for panels of >= 8 tiles (the wide early panels of a large factorisation)
 *   "dataflow_lds_kib" -1 auto (default) / 0 / KiB: LDS request of the 64-tile dataflow kernel; > 80 puts ONE workgroup on a
 *                  CU, which keeps the chain's latency-bound waves off SIMDs busy with another workgroup's MFMAs: auto
 *                  asks for 84 KiB while the launch has <= 2 700 tile tasks (one theta up to N ~ 4 600: -2..-7 %)
 *   "bcast_chunks" 0/1 (default 1): sharded evaluation -- a factored panel is broadcast one tile column at a time, each as soon as
 *                  it is final, instead of as one message after the whole panel (must agree on all ranks: checked)
 *   "fuse_potrf"   0/1 (default 1): calls of <= 8 thetas -- the panel-stream update that completes a diagonal tile also factors
 *                  it (no separate potrf128 launch, which waits 100-250 us for a CU slot under the trailing update): N = 16384 -3 %
 *   "dataflow_occ3" -1 auto (default) / 0 / 1: the 64-tile dataflow kernel in its three-workgroups-per-CU build (166 registers);
 *                  auto = launches of >= 6 000 tile tasks, which are throughput bound (N = 12288: -6.5 %)
 *   "fused_eval"   0/1 (default 1): a pure likelihood call of <= 8 thetas that qualifies for 64-tile dataflow
 *                  runs as ONE kernel launch (K(theta) tiles built inside the kernel, results written to
 *                  pinned host memory by its last task)
 *   "grad_potri"   0/1 gradient: form K^-1 = U U^T in one go when 2 N^2 of scratch fits (default 1), else
 *                  stream it in row blocks through forward + backward substitution
 *                  (1: where the factorisation is ONE dataflow launch -- N <= 12288 in fp64 -- U = L^-T for that contraction
 *                  comes from a second launch of the same kernel whose tasks are the tiles of U, and alpha = U z from one pass
 *                  over U, instead of ~5 dependent launches per tile column; 2: U always from the multi-kernel forward pass)
 *   "predict_df"   (default 2048; 0 = off): gphip_predict after a fit that was ONE dataflow launch runs the forward substitution
 *                  L^-1 k* of up to this many test points (twice that up to N = 8192) as one launch of the same kernel
 *                  (tasks = 64 x 64 tiles of the right-hand-side rows) instead of two launches per tile column; also after a
 *                  look-ahead-schedule fit up to "predict_df_max_nt" tile columns, and for gphip_predict_samples (all posterior
 *                  samples of a pass in ONE such launch, slot = sample, while the launch has <= 12 000 tasks)
 *   "thin_tiles"   0/1 (default 1): the GEMM kernel skips work whose result is known or never read -- all but the first
 *                  of the 128 bordered right-hand-side rows (zero), and the strictly-upper quadrant of diagonal tiles
 *   "max_slots"    cap on concurrently resident batch matrices
 *   "shard_min_n"  multi-device handles: shard ONE factorisation over the devices from this N on (default 16384;
 *                  0 = always, even for a world of one)
 *   "replicate_factor"  multi-device handles, sharded evaluations: 0 (default) every rank keeps ONLY its own outer panels
 *                  (compact storage + three receive buffers: ~1 / world of the workspace per rank); a fit leaves the factor
 *                  distributed and gphip_predict streams its panels through the ranks once more (a COLLECTIVE call for
 *                  rank handles in separate processes); 1: every rank keeps the dense workspace and receives panels in
 *                  place, so that after a fit all ranks hold all of L and prediction needs no further traffic
 *   "share_local_panels"  0/1 (default 1): ranks that share the owner's GPU (virtual ranks of a 1-GPU box, device-copy
 *                  communicator) read a factored panel where the owner keeps it instead of copying it; 0 forces the copies
 *                  through the rotating receive buffers (what distinct GPUs do)
 *   "bcast_two_hop"  -1 by size (default: on from 4 ranks when the loaded RCCL has send / recv / group calls) / 0 / 1: sharded
 *                  evaluation over RCCL with more than two ranks -- every panel message goes out as
 *                  scatter (the owner sends piece r to rank r: grouped ncclSend / ncclRecv) + in-place ncclAllGather instead of one
 *                  ncclBroadcast, so that all links of the xGMI mesh carry 1 / world of the message at once (never measured on
 *                  real multi-GPU hardware; results are bit-identical; must agree on all ranks: checked)
 *   "panel_df"     -1 by size (default) / 0 / 1: one theta, fp64, look-ahead schedule -- every outer panel (the look-ahead update by
 *                  the panel before it + its own factorisation) is ONE 64-tile dataflow launch whose tasks read the finished panel as
 *                  extra slabs; by size = 92 <= Nt <= 120 (N = 11k-15k: -2..-8 %, with an 80-column dataflow tail behind the panels)
 *   "dist_panel_df" -1 (default: 3 from 2 ranks on; where the device has no stream-ordered wait on memory, 2 at world 2, else 0)
 *                  / 0 / 1 / 2 / 3: sharded evaluation, fp64 -- the owner factors its outer panel as ONE 64-tile dataflow launch
 *                  (1), which also applies the look-ahead update, reading the previous panel from the receive buffer (2): the owner's
 *                  chain of kernels 31.8 -> 22.6 ms per N = 32768 evaluation with the chip to itself, but a panel is then final only
 *                  when its launch ends, so "bcast_chunks" cannot overlap its columns with the factorisation any more; (3) = (2)
 *                  with a counter per 128-wide tile column that the launch's tasks bump when they finish a tile of it: the
 *                  communication stream waits on the counter (hipStreamWaitValue32, value = the column's tile count) and sends
 *                  the column while the launch still works on the later ones.  Same arithmetic as 2 (bit-identical results).
 *                  bench.py --gpus N times every form.  Rank-local: need not agree across ranks.  "last_dist_panel_df"
 *                  (read-only): the form the last sharded evaluation used.
 *   "dist_owner_yield" -1 (default: on from 4 ranks) / 0 / 1: sharded evaluation -- on the rank that factors outer panel k + 1 the
 *                  remainder of its trailing update REST(k - 1) and its REST(k) are queued behind the END event of that panel
 *                  launch instead of sharing the GPU with it (only the piece of REST(k - 1) on panel k + 1 itself runs before):
 *                  a CU hands a freed slot to the next workgroup of the launch it is already dispatching whatever the stream
 *                  priorities, so a panel launch that becomes ready under a trailing GEMM of the same GPU otherwise runs at
 *                  that GEMM's pace (scripts/gpu_cu_partition.py: 8-10 ms instead of 1.2).  Results are bit-identical either way.
 *                  Rank-local.
 *   "last_issue_us" (read-only): host microseconds the last sharded evaluation spent issuing its schedule (all members of a
 *                  one-process group together).
 *   "trsv"         0 / 1 (default): gphip_solve of up to 4 right-hand sides (up to 16 from Nt >= 96, in batches of 4) and the
 *                  alpha = K^-1 (y - m) of gphip_fit run each triangle as ONE persistent launch that streams the factor once
 *                  (csrc/gp_trsv.h: tiles in registers, a band-2 chain of workgroup pairs, sentinel hand-offs); 0 = the
 *                  GEMM-shaped substitution for every count.  fp64 and fp32.
 *   "predict_df_max_nt" (default 256): gphip_predict / gphip_solve after a look-ahead fit (Nt above "dataflow_max_nt") still run
 *                  their forward / backward substitutions as single dataflow launches up to this many 128-tiles.
 *   "kbuild_mfma"  0 / 1 (default) / 2: the kernel-matrix build of the SE / Matern-5/2 kernels with the cross term of the squared
 *                  distances on the matrix pipe (kbuild_mfma_kernel): never / for every theta whose accuracy bound
 *                  B = sum_k (halfrange_k / l_k)^2 <= "kbuild_mfma_bound" (default 512; fp32: / 8) holds AND whose
 *                  conditioning keeps that entry error out of the likelihood: eps max(B, 64) (1 + k(x,x) / min nugget) <=
 *                  10^-"kbuild_mfma_digits" (default 9; fp32: 6 digits up) / always (tests).  Every other theta is built by the
 *                  direct-difference kernel, per theta of a batch.
 *   "custom_grad"  0/1 (default 1): gphip_loglik_grad of a run-time compiled covariance function through forward-mode dual
 *                  numbers (one factorisation); 0 = central differences.  "grad_analytic" (read-only) = 1 after a gradient call
 *                  that took the one-factorisation route.
 *   "debug_fail_alloc" / "debug_fail_hip" n: tests only -- the n-th device allocation of the next slot allocation / the n-th checked
 *                  HIP call of the next collective sequence fails (fault injection of the multi-process tests).  The names exist
 *                  only in a process started with GPHIP_TEST_HOOKS=1 in its environment ("unknown option" otherwise) and are
 *                  never taken from GPHIP_OPTIONS.
 *   "panel", "shard_min_n", "replicate_factor", "bcast_chunks", "bcast_two_hop" must have the same value on every rank of a
 *   multi-process job (checked by one small all-reduce at the start of every sharded evaluation: a mismatch fails the call on ALL
 *   ranks). */
int gphip_set_option(gphip_handle h, const char* name, double value);
int gphip_get_option(gphip_handle h, const char* name, double* value);
/* The environment variable GPHIP_OPTIONS="name=value,name=value" presets options for every handle the process
 * creates (for hosts that bind only the evaluation entry points, e.g. the LibraryLink shim). */

/* Per-kernel-class timing, measured with HIP events on the handle's stream while "profile"=1.
 * class: 0 kbuild, 1 potrf, 2 trsm, 3 gemm (in-panel), 4 gemm/syrk (trailing), 5 total eval, 6 prediction epilogue.
 * Returns accumulated milliseconds, launches, algorithmic flops and bytes since the last reset (class 4: flops =
 * m (m+1) nb per launch, SURVEY.md §8d -- not the tile-granular count the MFMA pipe executes). */
#define GPHIP_NCLASS 7
int gphip_get_profile(gphip_handle h, int cls, double* ms, double* launches, double* flops,
                      double* bytes);
int gphip_reset_profile(gphip_handle h);

/* ---- the per-rank compute STEPS of the multi-GPU 1-D block-cyclic Cholesky; no reference
 * equivalent (the reference never shards one factorisation).  A multi-device / per-rank handle (gphip_create
 * with ndev > 1, gphip_create_rank) drives these itself; they stay exported so that a host may run the schedule
 * with its own collective instead -- dist_cholesky.py does (torch.distributed: RCCL on GPUs, gloo in the CPU
 * schedule tests).  Outer panel j ("panel" option tile columns) belongs to rank j % world. ---- */
int gphip_set_streams(gphip_handle h, void* main_stream, void* panel_stream); /* adopt hipStream_t's */
int gphip_dist_num_panels(gphip_handle h, int* nouter);
int gphip_dist_panel_shape(gphip_handle h, int k, int64_t* rows, int64_t* cols);
int gphip_dist_begin(gphip_handle h, const double* theta, int p, int rank, int world);
/* packed_dev: device buffer of rows*cols elements of the handle's dtype (see gphip_dist_panel_shape) */
int gphip_dist_factor_panel(gphip_handle h, int k, void* packed_dev);
int gphip_dist_update(gphip_handle h, int k, const void* packed_dev, int j_first, int j_last,
                      int on_panel_stream);
int gphip_dist_end(gphip_handle h, double* logdet_partial, double* quad, int* info);

/* Device memory (bytes) local rank `member` of the handle holds for factor storage right now: dense workspace slots +
 * compact own-panel storage + receive buffers (diagnostics: a sharded evaluation keeps ~1 / world of the workspace). */
int gphip_factor_bytes(gphip_handle h, int member, double* bytes);

/* Block until all work queued on the handle's stream is complete. */
int gphip_sync(gphip_handle h);

const char* gphip_last_error(gphip_handle h);   /* owned by the library */
const char* gphip_version(void);
int gphip_device_count(int* n);

#ifdef __cplusplus
}
#endif
#endif /* GPHIP_H */
