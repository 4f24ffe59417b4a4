// gphip.hip -- C ABI (include/gphip.h) and host orchestration of the GP likelihood path.
//
// Per evaluation (BGP:297-305 closure = K1 -> K2 -> K3 -> K4 -> K6):
//   k_scale + kbuild      lower-triangle tiles of K(theta) + nugget, plus r^T as an extra row
//   two-level right-looking Cholesky with look-ahead:
//       for each outer panel of `panel` 128-tiles:
//           for each 128-tile column b in the panel:  potrf128(b) [L_bb and W_b = L_bb^-1];
//                                                     gemm_nt<2>: X <- X W_b^T below b;
//                                                     gemm_nt<1>(K=128) on the rest of the panel
//           gemm_nt<0>(K=panel*128) trailing SYRK on everything to the right (MFMA, dominant)
//   finalize              log det, quadratic form (bordered row), info
//   host epilogue         -1/2 (N log 2pi + logdet + quad)   (BGP:190-196)
// Work is queued on two handle-owned HIP streams (main + panel/look-ahead); X, y stay resident.
// Device arithmetic is fp64 (dtype 64) or fp32 (dtype 32); the ABI is fp64 either way.
#include "gp_kernels.h"
#include "gp_trsv.h"
#include "rccl_dyn.h"
#include "rtc_dyn.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gphip.h"

using namespace gphip;

namespace {

constexpr double LOG_TWO_PI = 1.8378770664093454835606594728112;
constexpr size_t GEMM_LDS = 2 * STAGE_BYTES;
// dTicket: [0] ticket, [1] abort word, [2 ..] DF_PARK_SLOTS ints of per-CU park counters, then the chain launch's ticket
constexpr size_t DF_TICKET_BYTES = (2 + DF_PARK_SLOTS * 4 / 8 + 2) * 8;

struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
    double flops, bytes;
};

}  // namespace

struct gphip_ctx {
    std::recursive_mutex mu;
    int device = 0;
    int dtype = 64;                    // 64: double, 32: float
    size_t es = 8;                     // element size of the device arithmetic type
    hipStream_t stream = nullptr;      // main stream: build, trailing updates, copies
    hipStream_t pstream = nullptr;     // panel stream (high priority): look-ahead panel factorisation
    hipStream_t cs = nullptr;          // stream the launch helpers currently target
    std::vector<hipEvent_t> sync_events;
    size_t sync_used = 0;
    int dist_first_factored = -1;                // sharded evaluation: outer panel whose first diagonal block the last LA update factored
    int bcast_chunks = 1;                        // sharded evaluation: a factored panel is broadcast one tile column at a time
    const void* df_prev_ptr = nullptr; int df_prev_k = -2;   // dist_panel_df = 2: where the previous outer panel lies (gphip_dist_update's deferred look-ahead step)
    static constexpr int df_panel_one_wg_tasks = 600;   // sharded schedule, dataflow panels: one workgroup per CU up to this many tile tasks
    // covariance function supplied as source text (gphip_create_custom): compiled at run time into the kernel build
    bool custom = false; int ncp = 0;            // ncp = its hyper-parameters p_0 .. p_{ncp-1}
    hipModule_t cmod = nullptr; hipFunction_t f_cbuild = nullptr, f_cdiag = nullptr, f_cprep = nullptr;
    // its gradient: the same text instantiated with forward-mode dual numbers (gp_dual.h), compiled at the first
    // gphip_loglik_grad (cgrad_state 0 = not tried yet, 1 = loaded, -1 = the text does not compile that way: differences)
    std::string custom_body, arch;
    hipModule_t cgmod = nullptr; hipFunction_t f_cgrad = nullptr;
    int cgrad_state = 0;
    int custom_grad = 1;                         // option: 0 = always central differences of the likelihood
    int grad_analytic = 0;                       // read-only: the last gphip_loglik_grad used the one-factorisation route
    double *dCustomP = nullptr, *hCustomP = nullptr;   // [slot][ncp]
    double* dKss = nullptr; size_t kss_cap = 0;  // k(x*, x*) of the current test points, [slot][mpad]
    int panel_df = -1;                           // one-GPU look-ahead schedule, one theta, fp64: every outer panel as ONE fused dataflow launch
    void* dW64s = nullptr; size_t w64s_bytes = 0;                  // gphip_predict_samples: 64-block inverses of every slot (samples_forward_df)
    double* dGpart = nullptr; size_t gpart_bytes = 0, ngacc = 0;   // gradient reduction: per-workgroup accumulator rows (grad_rows)
    int dist_owner_yield = -1;                   // sharded schedule: the owner's trailing updates wait for its panel launch (-1 = from 4 ranks)
    int dist_panel_df = -1;                      // sharded evaluation, fp64: the owner factors its outer panel as ONE 64-tile dataflow launch
                                                 // (1), which also applies the look-ahead update (2), and hands its tile columns to the
                                                 // broadcast stream by counters while it runs (3); -1 (default) = 3 from 2 ranks on where
                                                 // the device has stream-ordered waits on memory (gphip_dist_begin)
    bool dist_df_active = false;                 // the current sharded evaluation runs with dataflow panels (64-block partials / inverses)
    int dist_df_mode = 0;                        // .. and which form (dist_panel_df resolved: 0, 1, 2, 3; readable as option last_dist_panel_df)
    int bcast_two_hop = -1;                      // sharded evaluation over RCCL, world > 2: every broadcast as scatter (send / recv) + in-place all-gather;
                                                 // -1 (default) = on from 4 ranks when the loaded RCCL has send / recv / group calls (two_hop_ok)
    std::vector<hipEvent_t>* col_events = nullptr;   // queue_panel: record "tile column final" events here (owner of a sharded panel)
    // dist_panel_df = 3: a dataflow panel launch counts finished tiles per tile column (DfArgs::colsig); the owner's communication
    // stream waits for a column's count with hipStreamWaitValue32.  Counters are cumulative over the panels of one evaluation
    // (zeroed in gphip_dist_begin): the target of a wait = everything counted before + the column's own tiles.
    unsigned int* dColSig = nullptr;                 // [64] counters (the first `panel` are used)
    unsigned int colsig_target[64] = {0};
    std::vector<std::pair<unsigned int*, unsigned int>>* col_waits = nullptr;   // where gphip_dist_factor_panel reports (address, target) per tile column
    int fuse_potrf = 1;                          // option: panel-stream updates factor the diagonal tile they have just updated
    int fuse_b = -1;                             // launch_gemm: request (tile to factor) ...
    bool fuse_done = false;                      // ... and answer (the launch took it)
    int dataflow_occ3 = -1;                      // 64-tile kernel built for three workgroups per CU: -1 auto (>= 8 000 tasks), 0 never, 1 always
    static constexpr int dataflow_park = 1;      // 64-tile dataflow, two workgroups per CU: park the neighbour of a chain task (round 5 retune: never worse)
    int dataflow_lds_kib = -1;                   // LDS request of the 64-tile dataflow kernel (> 80: ONE workgroup per CU); -1 auto, 0 off
    bool own_streams = true;
    int dist_rank = 0, dist_world = 0;  // > 0 between gphip_dist_begin and gphip_dist_end
    bool dist_theta_ok = true;
    int64_t N = 0, d = 0, Npad = 0, Nt = 0;
    int64_t R = 0, slot_elems = 0;     // packed tile-major workspace: R = Nt + 1 tile rows, R (R + 1) / 2 tiles of 128 x 128 per slot
    int kernel_id = 0, mean_id = 0, nl = 0, p = 0, kt = 0;   // kt: 0 SE / 1 Matern-5/2 fast paths, 2 the general form (ks)
    // theta layout: [term 1: l_1..l_nl, (alpha), sf] [term 2: the same] [c] sn [mu]   (one plain term: l.., sf, sn[, mu])
    KSpec ks{0, 0, 0, 0};
    int nl2 = 0;                       // length scales of term 2 (0: no second term)
    bool has_a1 = false, has_a2 = false;
    void* dXs2 = nullptr;              // typed [slots][d][Npad]: inputs scaled by term 2's length scales
    void* dXsS2 = nullptr;             // typed [vcap][d]: test points scaled by term 2's length scales
    double *dInvEll2 = nullptr, *hInvEll2 = nullptr;
    double *dNullMu = nullptr, *dNullOut = nullptr;          // null-kernel path: per-theta mu and the two sums
    int null_cap = 0;
    void *dXt = nullptr, *dY = nullptr;                     // typed: [d][Npad], [Npad]
    double* dExp2 = nullptr;                                // [EXP_TAB] 2^(j/512): the kernel build's exp table
    // Kernel build with the distance cross term on the matrix pipe (kbuild_mfma_kernel): mid-range and half range of the
    // training inputs per dimension.  A theta's slot goes to that kernel while sum_k (half_k / l_k)^2 <= kbuild_mfma_bound
    // (fp32: / 8) -- the rounding error of its squared distances grows with that sum (gp_kernels.h); above it, and for every
    // other covariance form, kbuild_kernel builds the slot.  Option kbuild_mfma: 0 never, 1 by the bound, 2 always (tests).
    // The bound B limits the error of an ENTRY (eps B k_ij); what the likelihood sees of it is amplified by the
    // conditioning of K: with dK_ij = k_ij eta_ij, |eta| <= eps B, the quadratic form moves by a' dK a <= eps B sf^2 |a|^2
    // <= eps B (sf^2 / sn_min^2) r'K^-1 r and log det by tr(K^-1 dK) <= eps B sf^2 sqrt(N) / sn_min^2 against ~N.  So the
    // second half of the verdict is eps max(B, 64) (1 + k(x,x) / min nugget) <= 10^-kbuild_mfma_digits (fp64).  Calibrated on
    // clustered inputs (scripts/gpu_clustered_margin.py -> profiles/r06_clustered_margin.txt: N = 1500-6000, d = 1-3, B = 8-500,
    // sn = 1e-3 .. 3e-2, 360 cases): the two forms' likelihoods differ by at most 0.15 x this bound (the floor of 64: below
    // it the difference no longer shrinks with B -- table exponential, other summation order), so at the default 1e-9
    // a theta routed to the MFMA form stays within 1.5e-10 of the direct form.  The ill-conditioned theta (near-duplicate
    // points, small nugget) go to the direct form, whose error is eps r^2 k_ij.
    double* dCentre = nullptr;                              // [d]
    std::vector<double> x_centre, x_half;
    int kbuild_mfma = 1, kbuild_mfma_bound = 512, kbuild_mfma_digits = 9;
    double test_ratio = 0.0;                                // current test points: largest |x* - centre| / half range over the dimensions
    // batch workspace
    int slots = 0;
    void *dA = nullptr, *dXs = nullptr, *dW = nullptr;      // typed
    void* dDinv = nullptr;                                  // typed [slots][2 Nt][4][16 x 16]: DfArgs::D
    void* dW64 = nullptr;                                   // typed [2 Nt][64 x 64]: the 64-block inverses of a single-launch factorisation
                                                            // whose caller substitutes afterwards (fit, gradient); dW then takes the 128-blocks
    unsigned long w64_gen = ~0ul;                           // ws_gen of the factor dW64 belongs to
    void *dLT = nullptr, *dW64T = nullptr;                  // typed: the factor / the 64-block inverses with every 64 x 64 block transposed in place
    unsigned long lt_gen = ~0ul;                            // (gphip_solve's backward launch, DfArgs::LT); ws_gen of the factor they were made from
    double *dInvEll = nullptr, *dSlotp = nullptr, *dPartial = nullptr, *dRes = nullptr;
    int* dInfo = nullptr;
    double *hInvEll = nullptr, *hSlotp = nullptr, *hRes = nullptr;
    int* hInfo = nullptr;
    // options
    int panel = 4, profile = 0, swizzle = 1, max_slots = 256, lookahead = 1;
    int supertile = 2;                 // trailing SYRK tile order: 0 column-major chunks per XCD, 2 the tile LIST in 8x8 super-tile order,
                                       // equal chunks per XCD (3: for batches too)
    int latency_gemm = 1;                        // launches of <= latency_tiles tiles use the latency GEMM shape
    static constexpr int latency_tiles = 256;
    int latency_max_nt = 48;                     // ... for problems of at most this many tile columns (beyond: its 147 KB of LDS evicts trailing-SYRK workgroups)
    int dataflow = 1, dataflow_max_nt = 96;      // single-launch dataflow Cholesky: latency regime only
    int dataflow_max_slots = -1;                 // -1 (default): fp64 batches go through ONE dataflow launch by TASK COUNT (dataflow_max_tasks);
                                                 // n >= 1: the pre-round-6 rule -- up to n thetas (a few more of a small problem)
    int dataflow_max_tasks = 34000;              // 64-tile tasks of all slots together up to which one dataflow launch beats the multi-kernel batch
    int dataflow_fine_nt = 96;                   // ... with 64x64 tiles up to this many 128-tiles (fp64; measured best up to N = 12288)
    int panel_left = -1;                         // in-panel updates left-looking: -1 auto (batches), 0 never, 1 always
    int fuse_option = 1;                         // allow the single-launch evaluation (option "fused_eval")
    int grad_potri = 1;                          // gradient: K^-1 = U U^T in one go when the memory is there (1: U = L^-T from the dataflow kernel's
                                                 // inverse launch where the factor came from one launch; 2: always from the multi-kernel forward pass)
    int predict_df_max_nt = 256;                 // .. and, after a look-ahead-schedule fit, up to this many tile columns (N = 32768)
    int predict_df = 2048;                       // prediction after a single-launch fit: forward substitution as ONE dataflow launch up to this many (padded) test points
                                                 // (twice that up to N = 8192); 0 = never
    int panel_wide = 1;                          // wider outer panels while the trailing matrix is large (queue_factor)
    int thin_tiles = 1;                          // gemm_nt: skip the zero rows of the rhs block-row and the unread upper quadrant of diagonal tiles
    int debug_fail_alloc = 0;                    // tests: make the n-th device allocation of the next slot (re)allocation fail
    int dataflow_tail = 64;                      // large N: the last <= dataflow_tail tile columns go to the dataflow kernel (0 = off)
    bool fused_eval = false;                     // eval_chunk: the whole evaluation is ONE dataflow launch (build + factor + results)
    bool theta_packed = false;                   // eval_chunk: hyper-parameters travel as kernel arguments (k_scale_theta)
    bool want_w = false;                         // the caller substitutes with W_b afterwards (fit / predict / gradient)
    bool want_u = false, u_ready = false;        // gradient: a single-launch factorisation is followed by the inverse launch of the same
                                                 // kernel (U = L^-T into dV, see DfArgs::U); u_ready: it ran for the current factor
    int* dFlags = nullptr;                       // [slots][(Nt+1)^2] ready flags (value = epoch)
    unsigned long long* dTicket = nullptr;       // task ticket counter (+ abort flag in the next word)
    unsigned long long ticket_base = 0;
    int epoch = 0;
    // fitted state (slot 0)
    bool fitted = false;
    std::vector<double> theta_fit;
    // Fit epoch: ws_gen counts the calls that overwrote this context's workspace, fit_gen is ws_gen at the time the
    // resident factor was made, fit_id the number of the group's sharded fit that made it.  A factor is used only
    // while fit_gen == ws_gen (and, when test points shard over a group's members, while every member carries the
    // group's current fit_id) -- so a member's stale factor can never serve a later fit's prediction, whatever theta.
    unsigned long ws_gen = 0, fit_gen = ~0ul, fit_id = 0;
    double logdet_fit = 0, mu_fit = 0, kappa_fit = 0;
    // prediction / solve scratch
    void *dV = nullptr, *dXsT = nullptr, *dXsS = nullptr;   // typed
    double *dMean = nullptr, *dVar = nullptr;
    double* dPart = nullptr;                                 // strip partials of the prediction epilogue
    size_t part_cap = 0;
    int64_t vcap = 0;
    void* dAlpha = nullptr;                                  // typed [Npad] (gradient)
    // single-vector substitutions (gp_trsv.h): input block + two passes of {solution, chain copy, row sums, ticket} (trsv_pass_elems)
    void* dTrsvX = nullptr;
    void* dRows = nullptr; size_t rows_cap = 0;              // gphip_solve, few vectors on the GEMM path: [mc][Npad] staging block (rows_to_vblock_kernel)
    void* dTrsvP = nullptr;                                  // typed [2 directions][2 gaps][Nt][128 x 128]: the chain's products (trsv_prep_kernel)
    unsigned long trsvp_gen = ~0ul;                          // ws_gen of the factor they were made from
    int trsv = 1;                                            // option: gphip_solve with <= 4 right-hand sides and alpha through trsv_dataflow_kernel
    int ncu = 0;                                             // compute units of the device
    void* dKinv = nullptr;                                   // typed [(Npad + GRAD_LD_PAD) x Npad] lower tiles of K^-1 (gradient, potri route)
    double* dGacc = nullptr;                                 // [d + 2] gradient accumulators
    // profiling
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    double acc_ms[GPHIP_NCLASS] = {0}, acc_n[GPHIP_NCLASS] = {0}, acc_flops[GPHIP_NCLASS] = {0},
           acc_bytes[GPHIP_NCLASS] = {0};
    std::string err;
    // multi-device ("group") handle: see gphip_multi.inc.  Only the public handle owns a group; its peers are plain
    // contexts (one per further rank living in this process).
    struct gphip_group* group = nullptr;
    int shard_min_n = 16384;                     // one factorisation is sharded over the group's devices from this N on
    hipStream_t cstream = nullptr;               // communication stream (panel broadcasts), group members only
    void* packed[3] = {nullptr, nullptr, nullptr};   // rotating packed-panel buffers, group members only
    size_t packed_bytes = 0;
    double* dScal8 = nullptr;                    // 8 doubles for the scalar all-reduces (multi-process groups)
    void* drain_buf = nullptr;                   // scratch a draining member broadcasts through (group_drain_buf)
    size_t drain_bytes = 0;
    int last_issue_us = 0;                       // read-only: host microseconds the last sharded evaluation spent issuing its schedule
    int debug_fail_hip = 0;                      // tests: make the n-th checked HIP call of the next collective sequence fail
    int fit_rank = 0, fit_world = 0;             // the layout a distributed fit was made in
    bool in_group_call = false;                  // set on a member while the group handle runs a sharded call on it
    // Sharded evaluation (gphip_dist_*): where this rank keeps ITS outer panels.  replicate_factor = 0 (default): a
    // compact buffer holding only the owned panels (+ the corner tile on rank 0) -- memory per rank ~ 1 / world of the
    // workspace; 1: the dense workspace dA (every received panel is received in place: all ranks end up with all of L).
    int replicate_factor = 0;
    int share_local_panels = 1;                  // virtual ranks on the owner's GPU read a factored panel where the owner keeps it
    void* dOwn = nullptr;                        // typed compact own-panel storage
    size_t own_bytes = 0;
    void* dist_base = nullptr;                   // dOwn or dA: base of the storage the current sharded evaluation runs in
    std::vector<long> dist_adj;                  // [nouter + 1] tiles to add to a dense tile index of panel slot q (owned slots)
    long* dDistAdj = nullptr;                    // device copy; null while every entry is 0 (dense)
    int lay_rank = -1, lay_world = 0, lay_panel = 0, lay_full = -1;   // what dist_adj / dOwn were laid out for
    void* ws_override = nullptr;                 // tl<T>() / queue_panel address this base instead of dA (one owned panel)
    bool dist_fit = false;                       // the factor of theta_fit is spread over the ranks (owned panels only)
    void* dZ = nullptr;                          // typed [Npad]: z = L^-1 r gathered while the panels stream by (sharded prediction)
    bool z_vector = false;                       // the prediction epilogue reads z from dZ (set only inside predict_streamed)
    bool null_fit = false;                       // fitted state of a null-kernel handle (no factor: K = diag(sn^2))
    // Point-dependent nugget / mean of the CURRENT call (gphip_*_pw, BGP:37, 113, 300, 408): host rows [B][N] (training
    // points) and [S][M] (test points), null = the constant forms; device copies per workspace slot / prediction chunk
    const double *pw_mean_host = nullptr, *pw_nug_host = nullptr;
    const double *pw_mean_test = nullptr, *pw_nug_test = nullptr;
    long pw_test_stride = 0;                     // elements between the samples' rows of pw_*_test
    void *dPwMean = nullptr, *dPwNug = nullptr;  // typed [pw_cap][Npad]
    int pw_cap = 0;
    bool pw_mean_on = false, pw_nug_on = false;  // queue_build reads the device copies
    double *dPwMeanT = nullptr, *dPwNugT = nullptr;   // [vcap] test-point values of the current prediction chunk
    std::vector<double> null_diag, null_mean_test;    // fitted null kernel with a point-dependent nugget: the diagonal
    std::vector<double> fit_pw_mean, fit_pw_nug;      // the point-dependent arrays a DISTRIBUTED fit was made with (local refit)
};

namespace {

#define HIPCHK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            char buf_[512];                                                                \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                  \
            h->err = buf_;                                                                 \
            return GPHIP_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

// run f<double>(args) or f<float>(args) according to the handle's device type
#define DISPATCH(h, f, ...) ((h)->dtype == 64 ? f<double>(__VA_ARGS__) : f<float>(__VA_ARGS__))

int fail(gphip_ctx* h, int code, const char* msg) {
    if (h) h->err = msg;
    return code;
}

// every call that overwrites the workspace (or frees it) ends the life of the resident factor
void invalidate_fit(gphip_ctx* h) {
    h->fitted = false;
    h->dist_fit = false;
    ++h->ws_gen;
}
// ... and a successful fit stamps the factor with the workspace generation it lives in
void stamp_fit(gphip_ctx* h);
bool has_fit(const gphip_ctx* h) { return h->fitted && h->fit_gen == h->ws_gen; }

double pivot_tol_rel(const gphip_ctx* h) {
    return 64.0 * (h->dtype == 64 ? 2.220446049250313e-16 : 1.1920929e-07);
}

hipEvent_t get_event(gphip_ctx* h) {
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

// profile levels: 1 = kbuild + trailing SYRK + whole eval + prediction epilogue, 2 = every class
struct ProfScope {
    gphip_ctx* h;
    bool on;
    ProfRec r;
    ProfScope(gphip_ctx* h_, int cls, double flops, double bytes) : h(h_) {
        on = h->profile >= 2 || (h->profile == 1 && (cls >= 4 || cls == 0));   // 1: kbuild, trailing SYRK, totals, epilogue
        if (on) {
            r.cls = cls;
            r.flops = flops;
            r.bytes = bytes;
            r.e0 = get_event(h);
            r.e1 = get_event(h);
            (void)hipEventRecord(r.e0, h->cs);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(r.e1, h->cs);
            h->recs.push_back(r);
        }
    }
};

void harvest(gphip_ctx* h) {   // call after stream sync
    for (auto& r : h->recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        h->acc_ms[r.cls] += ms;
        h->acc_n[r.cls] += 1;
        h->acc_flops[r.cls] += r.flops;
        h->acc_bytes[r.cls] += r.bytes;
        h->pool.push_back(r.e0);
        h->pool.push_back(r.e1);
    }
    h->recs.clear();
}

void free_slots(gphip_ctx* h) {
    (void)hipFree(h->dA); (void)hipFree(h->dXs); (void)hipFree(h->dInvEll); (void)hipFree(h->dSlotp);
    (void)hipFree(h->dW); (void)hipFree(h->dPartial); (void)hipFree(h->dRes); (void)hipFree(h->dInfo);
    (void)hipFree(h->dFlags); (void)hipFree(h->dTicket); (void)hipFree(h->dW64); (void)hipFree(h->dDinv); (void)hipFree(h->dLT); (void)hipFree(h->dW64T);
    h->dW64 = h->dDinv = h->dLT = h->dW64T = nullptr; h->w64_gen = h->lt_gen = ~0ul;
    (void)hipFree(h->dPwMean); (void)hipFree(h->dPwNug);
    h->dPwMean = h->dPwNug = nullptr; h->pw_cap = 0;
    (void)hipFree(h->dXs2); (void)hipFree(h->dInvEll2); (void)hipHostFree(h->hInvEll2);
    h->dXs2 = nullptr; h->dInvEll2 = h->hInvEll2 = nullptr;
    (void)hipFree(h->dCustomP); (void)hipHostFree(h->hCustomP);
    h->dCustomP = h->hCustomP = nullptr;
    h->dFlags = nullptr; h->dTicket = nullptr; h->ticket_base = 0;
    (void)hipHostFree(h->hInvEll); (void)hipHostFree(h->hSlotp); (void)hipHostFree(h->hRes);
    (void)hipHostFree(h->hInfo);
    if (h->dist_base == h->dA) { h->dist_base = nullptr; if (h->lay_full == 1) { h->lay_rank = -1; h->lay_full = -1; } }
    h->dA = h->dXs = h->dW = nullptr;
    h->dInvEll = h->dSlotp = h->dPartial = h->dRes = nullptr;
    h->dInfo = nullptr;
    h->hInvEll = h->hSlotp = h->hRes = nullptr;
    h->hInfo = nullptr;
    h->slots = 0;
    invalidate_fit(h);                         // (a distributed fit keeps its block inverses / scalars in these buffers too)
}

size_t slot_bytes(const gphip_ctx* h) {
    return ((size_t)h->slot_elems + (size_t)(h->nl2 > 0 ? 2 : 1) * h->d * h->Npad + (size_t)h->Nt * TB * TB + (size_t)2 * h->Nt * 1024) * h->es +
           (size_t)h->Nt * 16 + (size_t)(2 * h->Nt + 1) * (2 * h->Nt + 1) * 4 + 4096;
}

// workspace = false (sharded evaluations of a rank that keeps only its own panels): everything a slot needs EXCEPT the
// workspace matrix itself (scaled inputs, block inverses, scalars, flags)
int ensure_slots(gphip_ctx* h, int want, bool workspace = true) {
    if (want <= h->slots && (h->dA || !workspace)) return GPHIP_OK;
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    const size_t per_slot = slot_bytes(h) - (workspace ? 0 : (size_t)h->slot_elems * h->es);
    fr += (size_t)h->slots * (slot_bytes(h) - (h->dA ? 0 : (size_t)h->slot_elems * h->es));   // what we are about to give back
    int fit = (int)((double)fr * 0.85 / (double)per_slot);
    if (fit < 1) return fail(h, GPHIP_ERR_HIP, "not enough device memory for one workspace matrix");
    if (want > fit) want = fit;
    if (want > h->max_slots) want = h->max_slots;
    if (want <= h->slots && (h->dA || !workspace)) return GPHIP_OK;
    if (want < h->slots) want = h->slots;                 // (adding the workspace to existing small slots)
    free_slots(h);
    const size_t S = (size_t)want;
    // Any failure below rolls back to the consistent ZERO-slot state (everything freed, slots = 0): the handle stays
    // usable and the next call simply allocates again (scripts/gpu_api_fuzz.py drives this to out-of-memory).
    hipError_t e = hipSuccess;
    const char* what = "";
    auto dev = [&](void** p, size_t bytes, const char* name) {
        if (e != hipSuccess) return;
        if (h->debug_fail_alloc > 0 && --h->debug_fail_alloc == 0) { e = hipErrorOutOfMemory; what = name; return; }   // fault injection (tests)
        if ((e = hipMalloc(p, bytes)) != hipSuccess) what = name;
    };
    auto host = [&](void** p, size_t bytes, const char* name) {
        if (e == hipSuccess && (e = hipHostMalloc(p, bytes)) != hipSuccess) what = name;
    };
    const size_t nflags = S * (size_t)(2 * h->Nt + 1) * (size_t)(2 * h->Nt + 1) * 4;
    if (workspace) dev(&h->dA, S * (size_t)h->slot_elems * h->es, "workspace");
    dev(&h->dXs, S * h->d * h->Npad * h->es, "scaled inputs");
    dev(&h->dW, S * h->Nt * TB * TB * h->es, "block inverses");
    dev(&h->dDinv, S * 2 * h->Nt * 1024 * h->es, "diagonal inverses of the 64-blocks");      // (64-tile dataflow chain, DfArgs::D)
    dev((void**)&h->dInvEll, S * h->d * 8, "inverse length scales");
    if (h->nl2 > 0) {                          // second term of a sum / product kernel: its own scaled copy of the inputs
        dev(&h->dXs2, S * h->d * h->Npad * h->es, "scaled inputs (term 2)");
        dev((void**)&h->dInvEll2, S * h->d * 8, "inverse length scales (term 2)");
        host((void**)&h->hInvEll2, S * h->d * 8, "pinned inverse length scales (term 2)");
    }
    dev((void**)&h->dSlotp, S * SLOTP * 8, "slot scalars");
    dev((void**)&h->dPartial, S * 2 * h->Nt * 8, "log-det partials");     // per 64-block in the fine dataflow schedule
    dev((void**)&h->dRes, S * 2 * 8, "results");
    dev((void**)&h->dInfo, S * 4, "info words");
    dev((void**)&h->dFlags, nflags, "dependency flags");
    dev((void**)&h->dTicket, DF_TICKET_BYTES, "ticket counter");        // + the per-CU "chain task here" counters
    host((void**)&h->hInvEll, S * h->d * 8, "pinned inverse length scales");
    host((void**)&h->hSlotp, S * SLOTP * 8, "pinned slot scalars");
    if (h->custom) {
        dev((void**)&h->dCustomP, S * (size_t)std::max(h->ncp, 1) * 8, "covariance-function parameters");
        host((void**)&h->hCustomP, S * (size_t)std::max(h->ncp, 1) * 8, "pinned covariance-function parameters");
    }
    host((void**)&h->hRes, S * 2 * 8, "pinned results");
    host((void**)&h->hInfo, (S + 2) * 4, "pinned info words");            // + the dataflow abort flag (factorisations: [nb]; later launches: [S + 1])
    // ON THE HANDLE'S STREAM: the handle's streams are non-blocking, so a null-stream hipMemset is not ordered before
    // the kernels queued next -- a dataflow task could read a recycled allocation's stale flags (another handle's epoch
    // numbers) or tickets before the clear landed.  Found by scripts/gpu_api_fuzz.py (wrong likelihood / memory fault
    // right after a batch grew the slot count), present since round 1.
    if (e == hipSuccess && (e = hipMemsetAsync(h->dFlags, 0, nflags, h->stream)) != hipSuccess) what = "flag clear";
    if (e == hipSuccess && (e = hipMemsetAsync(h->dTicket, 0, DF_TICKET_BYTES, h->stream)) != hipSuccess) what = "ticket clear";
    if (e != hipSuccess) {
        (void)hipGetLastError();               // (clear the sticky out-of-memory status)
        free_slots(h);
        char buf[256];
        snprintf(buf, sizeof buf, "allocating %d workspace slot(s) failed at '%s': %s", want, what, hipGetErrorString(e));
        h->err = buf;
        return GPHIP_ERR_HIP;
    }
    h->slots = want;
    return GPHIP_OK;
}

// ------------------------------------------------------------------------------------------
// typed launch helpers
// ------------------------------------------------------------------------------------------
template <typename T, int KT>
void launch_kbuild_kt(gphip_ctx* h, const KBuildArgs<T>& a, dim3 grid) {
    const int d = a.d;
    const size_t etab = sizeof(T) == 8 ? (size_t)EXP_TAB * 8 : 0;      // fp64: the sf2 2^(j/512) table behind the point tiles
#define KB_CASE(DD)                                                                                 \
    case DD:                                                                                        \
        hipLaunchKernelGGL((kbuild_kernel<T, DD, KT>), grid, dim3(256), (size_t)DD * TB * sizeof(T) + etab, h->cs, a); \
        break;
    switch (d) {
        KB_CASE(1) KB_CASE(2) KB_CASE(3) KB_CASE(4) KB_CASE(5) KB_CASE(6) KB_CASE(7) KB_CASE(8)
        KB_CASE(16)
        default:
            hipLaunchKernelGGL((kbuild_kernel<T, 0, KT>), grid, dim3(256), (d > KB_LDS_MAXD ? 0 : (size_t)2 * d * TB * sizeof(T)) + etab, h->cs, a);
    }
#undef KB_CASE
}

template <typename T, int KT>
void launch_kbuild_mfma_kt(gphip_ctx* h, const KBuildMArgs<T>& m, dim3 grid) {
    const int d = m.b.d;
#define KM_CASE(KS)                                                                                                  \
    case KS:                                                                                                         \
        hipLaunchKernelGGL((kbuild_mfma_kernel<T, KS, KT>), grid, dim3(256), (kbuild_mfma_lds<T, KS>(d)), h->cs, m); \
        break;
    switch ((d + 3) / 4) {
        KM_CASE(1) KM_CASE(2) KM_CASE(3) KM_CASE(4)
        default:
            hipLaunchKernelGGL((kbuild_mfma_kernel<T, 0, KT>), grid, dim3(256), (kbuild_mfma_lds<T, 0>(d)), h->cs, m);
    }
#undef KM_CASE
}

// Which family of kbuild_mfma_kernel serves this handle: the two fast paths (kt 0 / 1), and the general form when it is ONE
// term, with or without a constant offset -- SE (0), Matern-5/2 (1), Matern-3/2 (2), rational quadratic (3); -1: two-term kernels,
// run-time compiled functions.
int mfma_family(const gphip_ctx* h) {
    if (h->custom) return -1;
    if (h->kt <= 1) return h->kt;
    // (a constant offset c + k1 -- the reference's own example kernel, #2 + Exp[-(pt1 - pt2)^2 / #1^2], BGP:16 -- rides along: the
    //  kernel adds slot scalar SP_OFFSET, which is 0 without one)
    if (h->ks.op == 0 && h->nl2 == 0 && h->ks.fam1 >= 0 && h->ks.fam1 <= 3) return h->ks.fam1;
    return -1;
}

// xri / xrj: the RAW (unscaled) row / column points kbuild_mfma_kernel scales itself; null = this build has no such form
// (the direct kernel builds every slot)
template <typename T>
void launch_kbuild(gphip_ctx* h, const KBuildArgs<T>& a0, dim3 grid, const T* xri = nullptr, const T* xrj = nullptr) {
    const int mk = mfma_family(h);                 // family the matrix-pipe build serves this handle with (-1: none)
    if (!h->custom && mk >= 0 && xri && xrj) {
        int nm = 0;                                // slots of this launch the staged thetas hand to the matrix-pipe build
        for (unsigned s2 = 0; s2 < grid.y; ++s2) nm += h->hSlotp[(size_t)s2 * SLOTP + SP_MFMA] != 0.0;
        if (nm > 0) {
            KBuildMArgs<T> m{};
            m.b = a0; m.xri = xri; m.xrj = xrj; m.inv_ell = h->dInvEll; m.centre = h->dCentre;
            if (mk == 0) launch_kbuild_mfma_kt<T, 0>(h, m, grid);
            else if (mk == 1) launch_kbuild_mfma_kt<T, 1>(h, m, grid);
            else if (mk == 2) launch_kbuild_mfma_kt<T, 2>(h, m, grid);
            else launch_kbuild_mfma_kt<T, 3>(h, m, grid);
            if (nm == (int)grid.y) return;
        }
        KBuildArgs<T> a = a0;
        a.mfma_skip = nm > 0;
        if (h->kt == 0) launch_kbuild_kt<T, 0>(h, a, grid);
        else if (h->kt == 1) launch_kbuild_kt<T, 1>(h, a, grid);
        else hipLaunchKernelGGL((kbuild_kernel<T, 0, 2>), grid, dim3(256), a.d > KB_LDS_MAXD ? 64 : (size_t)4 * a.d * TB * sizeof(T), h->cs, a);
        return;
    }
    if (h->custom) {                               // the run-time compiled instantiation kbuild_kernel<T, 0, 3>
        KBuildArgs<T> a = a0;
        a.cp = h->dCustomP; a.ncp = std::max(h->ncp, 1);
        void* params[] = {&a};
        const size_t lds = (a.d > KB_LDS_MAXD ? 0 : (size_t)2 * a.d * TB * sizeof(T)) + (size_t)a.ncp * sizeof(double);   // point tiles + hyper-parameters
        (void)hipModuleLaunchKernel(h->f_cbuild, grid.x, grid.y, grid.z, 256, 1, 1, (unsigned)lds, h->cs, params, nullptr);
        return;
    }
    const KBuildArgs<T>& a = a0;
    if (h->kt == 0) launch_kbuild_kt<T, 0>(h, a, grid);
    else if (h->kt == 1) launch_kbuild_kt<T, 1>(h, a, grid);
    else        // general form: row and column points of both terms in LDS
        hipLaunchKernelGGL((kbuild_kernel<T, 0, 2>), grid, dim3(256), a.d > KB_LDS_MAXD ? 64 : (size_t)4 * a.d * TB * sizeof(T), h->cs, a);
}

// Outer panel boundaries of the multi-kernel factorisation.  Far from the end the trailing update is long and hides a wider
// panel's factorisation behind it, and a wider panel means fewer read-modify-write passes over the trailing matrix (measured,
// one theta, fp64: N=32768 190.2 -> 187.4 ms, N=49152 621 -> 610 ms; at N <= 16384 the base width is best): twice the base
// width while >= 192 tile columns remain, 1.5x while >= 128 remain (option "panel_wide", default 1).
std::vector<int> panel_bounds(const gphip_ctx* h) {
    const int Nt = (int)h->Nt, P = h->panel;
    std::vector<int> bnd{0};
    while (bnd.back() < Nt) {
        const int rem = Nt - bnd.back();
        int w = P;
        if (h->panel_wide) w = rem >= 192 ? 2 * P : (rem >= 128 ? P + P / 2 : P);
        bnd.push_back(std::min(Nt, bnd.back() + w));
    }
    return bnd;
}
bool use_dataflow(const gphip_ctx* h, int nslots);
inline int few_slots(const gphip_ctx* h) { return h->dataflow_max_slots < 0 ? 8 : h->dataflow_max_slots; }   // "a few thetas": schedule choices made for latency
hipEvent_t sync_event(gphip_ctx* h);

// queue k_scale + kbuild for nslots slots (theta already staged in dInvEll / dSlotp)
template <typename T>
int queue_build(gphip_ctx* h, int nslots) {
    const long tot = (long)h->d * h->Npad;
    int gx = (int)((tot + 255) / 256);
    if (gx > 1024) gx = 1024;
    if (h->theta_packed) {
        ThetaPack tp;
        memcpy(tp.v, h->hInvEll, (size_t)nslots * h->d * 8);
        memcpy(tp.v + (size_t)nslots * h->d, h->hSlotp, (size_t)nslots * SLOTP * 8);
        hipLaunchKernelGGL(k_scale_theta<T>, dim3(gx, nslots), dim3(256), 0, h->cs, (const T*)h->dXt, (T*)h->dXs, tp,
                           h->dInvEll, h->dSlotp, h->dInfo, (int)h->d, (int)h->Npad, nslots);
    } else {
        hipLaunchKernelGGL(k_scale<T>, dim3(gx, nslots), dim3(256), 0, h->cs, (const T*)h->dXt, (T*)h->dXs,
                           h->dInvEll, (int)h->d, (int)h->Npad);
    }
    if (h->nl2 > 0)
        hipLaunchKernelGGL(k_scale<T>, dim3(gx, nslots), dim3(256), 0, h->cs, (const T*)h->dXt, (T*)h->dXs2, h->dInvEll2,
                           (int)h->d, (int)h->Npad);
    if (h->custom) {                               // prior variance scale -> pivot tolerance, per slot (device: only it can evaluate k)
        const void* x = h->dXt;
        int npad = (int)h->Npad, n = (int)h->N, d = (int)h->d, ncp = std::max(h->ncp, 1);
        const double* cp = h->dCustomP;
        double* sp = h->dSlotp;
        void* params[] = {&x, &npad, &n, &d, &cp, &ncp, &sp};
        (void)hipModuleLaunchKernel(h->f_cprep, (unsigned)nslots, 1, 1, 256, 1, 1, 0, h->cs, params, nullptr);
    }
    KBuildArgs<T> a{};
    a.ks = h->ks; a.xi2 = a.xj2 = (const T*)h->dXs2;
    a.out = (T*)h->dA; a.ld = TB; a.bstride = h->slot_elems;
    a.xi = (const T*)h->dXs; a.xj = (const T*)h->dXs; a.xi_bstride = a.xj_bstride = tot;
    a.npad_i = a.npad_j = (int)h->Npad; a.n_i = a.n_j = (int)h->N;
    a.y = (const T*)h->dY; a.slotp = h->dSlotp; a.d = (int)h->d; a.mode = 0; a.exp2tab = h->dExp2;
    a.nt_i = (int)h->Nt + 1; a.nt_j = (int)h->Nt;
    a.own_panel = h->panel; a.own_world = h->dist_world; a.own_rank = h->dist_rank;
    if (h->dist_world > 0) { a.out = (T*)h->dist_base; a.adj = h->dDistAdj; }     // sharded evaluation: this rank's own storage
    a.pw_nug = h->pw_nug_on ? (const T*)h->dPwNug : nullptr;
    a.pw_mean = h->pw_mean_on ? (const T*)h->dPwMean : nullptr;
    a.pw_bstride = h->Npad;
    const long ntiles = (long)(h->Nt + 1) * (h->Nt + 2) / 2;
    ProfScope ps(h, 0, 0.0, (double)sizeof(T) * nslots * ((double)h->N * (h->N + 1) / 2 + (double)h->N * h->d));
    launch_kbuild<T>(h, a, dim3((unsigned)ntiles, nslots), (const T*)h->dXt, (const T*)h->dXt);
    return 0;
}

// cls: profile class (2 panel solve, 3 in-panel/look-ahead GEMM, 4 trailing SYRK, 6 = "NN" role)
// GEMM operand / result: a column-major block (R = 0: pointer, leading dimension, elements between slots) or the
// packed tile-major workspace (R = tile rows; p = slot-0 base, possibly shifted for a panel held outside the
// workspace; k0 = tile column at which the operand panel starts)
template <typename T>
struct Opnd {
    const T* p; long ld; long bs; int R; int k0;
    const long* adj = nullptr; int adj_panel = 0;      // C only: compact own-panel storage of a rank (GemmArgs::c_adj)
    int lower = 0;                                     // J operand only: lower-triangular block with explicit zeros above (W_b)
};
template <typename T>
Opnd<T> cm(const T* p, long ld, long bs) { return Opnd<T>{p, ld, bs, 0, 0}; }
// the 128-block inverse W_b = L_bb^-1 (lower triangular, explicit zeros above the diagonal) as the J operand of a solve
template <typename T>
Opnd<T> wb(const T* W, int b, long lrs) {
    Opnd<T> o{W + (long)b * TB * TB - (long)b * TB, TB, lrs, 0, 0};
    o.lower = 1;
    return o;
}
template <typename T>
Opnd<T> tl(const gphip_ctx* h, int k0 = 0, bool all_slots = true) {
    return Opnd<T>{(const T*)(h->ws_override ? h->ws_override : h->dA), TB, all_slots ? (long)h->slot_elems : 0l, (int)h->R, k0};
}

template <typename T>
size_t potrf_lds();

template <typename T>
void launch_gemm(gphip_ctx* h, int cls, Opnd<T> Co, Opnd<T> Ao, Opnd<T> Bo, int K, int r0, int r1, int c0, int c1, int tri,
                 int nslots, int mode = 0, int ktri = 0, int thin_row = -1, int groups = 1, int grp_stride = 0,
                 int grp_width = 0) {
    GemmArgs<T> g{};
    g.grp_stride = groups > 1 ? grp_stride : 0;
    g.grp_width = grp_width;
    g.grp_count = groups;
    g.mode = mode;
    g.ktri = ktri;
    // factorisation launches only (thin_row given): the bordered rhs block-row carries ONE real row, and nothing reads
    // the strictly-upper quadrant of a diagonal tile -- both are skipped inside the kernel (option "thin_tiles")
    g.thin_row = h->thin_tiles ? thin_row : -1;
    g.skip_upper = (h->thin_tiles && thin_row >= 0 && tri && mode == 0 && !ktri) ? 1 : 0;
    g.C = const_cast<T*>(Co.p); g.ldc = Co.ld; g.c_bstride = Co.bs; g.c_R = Co.R;
    g.c_adj = Co.adj; g.c_adj_panel = Co.adj_panel;
    g.b_lower = Bo.lower;
    g.A = Ao.p; g.lda = Ao.ld; g.a_bstride = Ao.bs; g.a_R = Ao.R; g.a_k0 = Ao.k0;
    g.B = Bo.p; g.ldb = Bo.ld; g.b_bstride = Bo.bs; g.b_R = Bo.R; g.b_k0 = Bo.k0;
    g.K = K; g.r0 = r0; g.r1 = r1; g.c0 = c0; g.c1 = c1; g.tri = tri;
    const int H = r1 - r0, W = c1 - c0;
    if (H <= 0 || W <= 0) return;
    if (!tri) {
        g.nrect = H * W;
        g.ntiles = H * W;
    } else {
        int nrc = r0 - c0;                      // columns left of the triangle: full height
        if (nrc > W) nrc = W;
        const int ntc = W - nrc;                // triangle columns (heights H, H-1, ..)
        g.nrect = nrc * H;
        g.ntiles = g.nrect + ntc * H - ntc * (ntc - 1) / 2;
    }
    // (not for ktri launches: their tiles contract k >= 128 ti only, so equal COUNTS of the column-major list are unequal work --
    //  the first XCD's chunk holds 17 % of it at N = 8192; dealt round-robin the XCDs finish together)
    g.swizzle = h->swizzle && g.ntiles >= 64 && !ktri;
    int grid_x = g.ntiles;
    if (tri && r0 == c0 && W == H && h->supertile >= 2 && H >= 16 && mode == 0 && groups == 1 && (nslots == 1 || h->supertile == 3)) {
        g.super = 2;                            // the tile list in blocked (8 x 8 super-tile) order, equal chunks per XCD
    }
    double flops = 2.0 * TB * TB * (double)K * g.ntiles * nslots;       // tile-granular (what the MFMA pipe executes)
    if (g.grp_stride > 0) {
        // grouped triangular launch: `groups` panels of width grp_width starting at c0, c0 + stride, .. (clipped to c1), each
        // from its own diagonal down to r1; grid.x = the first panel's tile count
        double fl = 0.0;
        long tiles = 0;
        for (int q = 0; q < groups; ++q) {
            const int a0 = c0 + q * grp_stride, a1 = std::min(a0 + grp_width, c1);
            const int Hq = r1 - a0, wq = a1 - a0;
            tiles += (long)wq * Hq - (long)wq * (wq - 1) / 2;
            const double a = (double)a0 * TB, b = std::min((double)a1 * TB, (double)h->N);
            const double cnt = b > a ? b - a : 0.0;
            fl += 2.0 * (double)K * (cnt * (double)h->N - (a + b - 1.0) * cnt / 2.0);
        }
        // dense grid: every group at full width (only the last one can be clipped: its surplus workgroups exit)
        long dense = 0;
        for (int q = 0; q < groups; ++q) {
            const int a0 = c0 + q * grp_stride, wq = std::min(grp_width, c1 - a0), Hq = r1 - a0;
            dense += (long)wq * Hq - (long)wq * (wq - 1) / 2;            // (only the last group can be narrower)
        }
        g.nrect = 0;
        g.ntiles = (int)dense;
        flops = cls == 4 ? fl : 2.0 * TB * TB * (double)K * (double)tiles;
    } else if (cls == 4 && tri && r0 == c0) {
        // trailing SYRK: report ALGORITHMIC flops (m (m+1) nb for a trailing matrix of m true
        // columns and a panel of width nb) -- full diagonal tiles, identity padding and the bordered rhs block-row
        // are executed but not counted.  General form for the tile columns [c0, c1) of an N-column matrix:
        // 2 K sum_{j = a}^{b-1} (N - j),  a = 128 c0, b = min(128 c1, N)
        const double a = (double)c0 * TB, b = std::min((double)c1 * TB, (double)h->N);
        const double cnt = b > a ? b - a : 0.0;
        flops = 2.0 * (double)K * (cnt * (double)h->N - (a + b - 1.0) * cnt / 2.0) * nslots;
    }
    // algorithmic bytes: C tiles read + written once, each operand panel streamed once
    const double bytes = (double)sizeof(T) * nslots * (2.0 * TB * TB * g.ntiles + (double)(tri ? H : H + W) * TB * K);
    ProfScope ps(h, cls == 6 ? 3 : cls, flops, bytes);
    const dim3 grid(g.grp_stride > 0 ? (unsigned)g.ntiles : (unsigned)grid_x, g.grp_stride > 0 ? 1u : (unsigned)nslots);
    // shape: 2x2 waves / 2 LDS stages (throughput, 2 workgroups per CU) or, for launches with at most one
    // tile per CU, 4x4 waves / 4 LDS stages with counted DMA waits (latency)
    // (measured: -8 % per evaluation at N=4096, neutral at 8192, +5 % at 32768 where its 147 KB of LDS keeps
    //  trailing-SYRK workgroups off the CU -- so it is used for small problems only)
    const bool lat = h->latency_gemm && !g.super && h->Nt <= h->latency_max_nt && (long)grid_x * nslots <= h->latency_tiles;
    // a panel-stream update asked to factor the diagonal tile it updates (queue_panel / queue_factor): 256-thread shape only
    g.fuse_b = -1;
    h->fuse_done = false;
    size_t lds2 = GEMM_LDS;
    if (h->fuse_b >= 0 && cls == 3 && mode == 0 && !lat && !ktri && groups == 1 && tri && h->fuse_b >= c0 && h->fuse_b < c1 &&
        h->fuse_b >= r0) {
        g.fuse_b = h->fuse_b;
        g.fuse_W = (T*)h->dW; g.fuse_partial = h->dPartial; g.fuse_info = h->dInfo; g.fuse_slotp = h->dSlotp; g.fuse_nt = (int)h->Nt;
        lds2 = std::max(lds2, potrf_lds<T>());
        h->fuse_done = true;
    }
    h->fuse_b = -1;
#define GEMM_LAUNCH(ROLE)                                                                                          \
    do {                                                                                                           \
        if (lat) hipLaunchKernelGGL((gemm_nt_kernel<T, ROLE, 4, 4, 4>), grid, dim3(1024), 2 * GEMM_LDS, h->cs, g);  \
        else hipLaunchKernelGGL((gemm_nt_kernel<T, ROLE, 2, 2, 2>), grid, dim3(256), GEMM_LDS, h->cs, g);           \
    } while (0)
    if (cls == 6) GEMM_LAUNCH(3);
    else if (mode == 1) GEMM_LAUNCH(2);
    else if (cls == 4) GEMM_LAUNCH(0);
    else if (g.fuse_b >= 0) hipLaunchKernelGGL((gemm_nt_kernel<T, 4, 2, 2, 2>), grid, dim3(256), lds2, h->cs, g);
    else GEMM_LAUNCH(1);
#undef GEMM_LAUNCH
}

hipEvent_t sync_event(gphip_ctx* h) {       // untimed events for cross-stream ordering
    if (h->sync_used == h->sync_events.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;   // (recording it then fails loudly)
        h->sync_events.push_back(e);
    }
    return h->sync_events[h->sync_used++];
}

template <typename T>
size_t potrf_lds() { return 16 + (size_t)PT_LDS_ELEMS * sizeof(T); }

// factor the tile columns [K0, K0+nin) of all slots (diagonal blocks, panel solves, in-panel updates)
template <typename T>
int queue_panel(gphip_ctx* h, int K0, int nin, int nslots, bool first_factored = false) {
    const int Nt = (int)h->Nt, R = Nt + 1;
    const long bs = h->slot_elems, lrs = (long)Nt * TB * TB;
    T* A = (T*)(h->ws_override ? h->ws_override : h->dA);
    T* W = (T*)h->dW;
    // In-panel updates: right-looking (after column b, K = 128 onto every remaining column of the panel: few,
    // wide launches -- shortest chain for one theta) or left-looking (before column b, ONE update of that
    // column with K = 128 (b - K0): each tile read and written once per panel, longer contractions --
    // better throughput once a batch of thetas fills the chip anyway).
    // auto: batches, and the wide (>= 8-tile) early panels of a large single factorisation (N = 32768: -0.7 %, N = 49152:
    // -0.8 %; with 4-6-tile panels right-looking is 1 % faster)
    const bool left = h->panel_left > 0 || (h->panel_left < 0 && (nslots > few_slots(h) || nin >= 8));
    // "fuse_potrf": an update launch that completes diagonal tile (b, b) also factors it (GemmArgs::fuse_b) -- then there is
    // no potrf128 launch for column b.  first_factored: the caller's look-ahead update already did that for column K0.
    // One theta (or a few): the diagonal block is latency-critical.  Batches are throughput bound and the fused kernel's 272
    // registers halve the resident waves of what are then BIG update launches (200 x N=4096: 86 -> 91 ms): not there.
    const bool fuse = h->fuse_potrf && nslots <= 8;
    bool factored = first_factored;
    for (int s = 0; s < nin; ++s) {
        const int b = K0 + s;
        if (left && s > 0) {
            if (fuse) h->fuse_b = b;
            launch_gemm<T>(h, 3, tl<T>(h), tl<T>(h, K0), tl<T>(h, K0), s * TB, b, R, b, b + 1, 1, nslots, 0, 0, Nt);
            factored = h->fuse_done;
        }
        if (!factored) {
            ProfScope ps(h, 1, 2.0 * TB * TB * TB / 3.0 * nslots, 0.0);
            hipLaunchKernelGGL(potrf128_kernel<T>, dim3(nslots), dim3(256), potrf_lds<T>(), h->cs, A, bs, b, W,
                               h->dPartial, Nt, h->dInfo, h->dSlotp);
        }
        // panel solve X <- X W_b^T for every row tile below the diagonal block (incl. rhs rows)
        launch_gemm<T>(h, 2, tl<T>(h), tl<T>(h, b), wb<T>(W, b, lrs), TB, b + 1, R, b, b + 1, 0, nslots, 1, 0, Nt);
        factored = false;
        if (h->col_events) {                   // tile column b is final from here on (later columns only read it)
            hipEvent_t e = sync_event(h);
            HIPCHK(hipEventRecord(e, h->cs));
            h->col_events->push_back(e);
        }
        if (!left && s + 1 < nin) {
            if (fuse) h->fuse_b = b + 1;
            launch_gemm<T>(h, 3, tl<T>(h), tl<T>(h, b), tl<T>(h, b), TB, b + 1, R, b + 1, K0 + nin, 1, nslots, 0, 0, Nt);
            factored = h->fuse_done;
        }
    }
    return 0;
}

// Single-launch dataflow Cholesky (chol_dataflow_kernel): one workgroup per tile, flags instead of
// launches.  Used for small / mid problems where the multi-kernel schedule is latency bound.
// 64x64 tiles (fp64, Nt <= dataflow_fine_nt) halve the serial chain per column once more; they leave
// inverses of 64-blocks in dW, so callers that substitute afterwards (h->want_w) get the 128-block
// inverses rebuilt by trtri128.
// Option "panel_df" of the look-ahead schedule (queue_factor): one theta, fp64 -- every outer panel (look-ahead update by the
// panel before it + its own factorisation) is ONE fused 64-tile dataflow launch.  -1 = by size: a few such panels, their
// trailing updates on the 128-tile GEMM, in front of an 80-column dataflow tail beat both the single launch and the
// multi-kernel panels for N = 11k-15k (round 4, profiles/r04_panel_df_sweep.txt: N=12288 13.2 -> 12.1 ms, 13312 15.7 -> 15.0,
// 14336 18.9 -> 18.4; no gain at N <= 10240, a loss from N = 16384 on, where the panel launches starve behind the
// trailing update's one-workgroup-per-CU GEMM).
bool panel_df_on(const gphip_ctx* h, int nslots) {
    if (h->dtype != 64 || nslots != 1 || !h->dataflow || !h->lookahead || h->dist_world > 0) return false;
    if ((h->Nt + h->panel - 1) / h->panel < 2) return false;
    if (h->panel_df >= 0) return h->panel_df != 0;
    return h->Nt >= 92 && h->Nt <= 120;
}

bool use_dataflow(const gphip_ctx* h, int nslots) {
    // (a fit or a gradient call prefers the single launch where it is allowed at all: its 64-block inverses serve the forward /
    //  inverse launches that follow, launch_dataflow_inverse)
    if (panel_df_on(h, nslots) && !((h->want_u || h->want_w) && h->Nt <= h->dataflow_max_nt)) return false;
    // measured: wins 1.05-2.3x for one theta up to N = 12288, ties at 8-16 slots, loses 2x at 200 slots
    // (there the multi-kernel schedule's big launches are throughput bound, not latency bound)
    if (!h->dataflow || h->dist_world > 0 || h->Nt > h->dataflow_max_nt) return false;
    // Round 6, re-measured with the round-5 kernel (scripts/gpu_batch_crossover.py, profiles/r06_batch_crossover.txt): for every
    // N = 512 .. 12288 the two schedules cross where ALL slots together have ~34-36 thousand 64-tile tasks -- 750 thetas at
    // N = 512, 128 at N = 1024, 16 at N = 4096, 4 at N = 8192, 2 at N = 12288.  The old rule (<= 8 thetas at any size, a few
    // more of a small problem) left up to 2.6x on the table for 9-100 thetas of N <= 2048 and lost 9-16 % for 6-8 thetas at
    // N >= 8192.
    if (h->dataflow_max_slots < 0 && h->dtype == 64 && h->Nt <= h->dataflow_fine_nt) {
        const long t64 = (long)(2 * h->Nt + 1) * (2 * h->Nt + 2) / 2 * nslots;
        return nslots == 1 || t64 <= h->dataflow_max_tasks;
    }
    // fp32 runs 128-tiles: the same sweep (profiles/r06_batch_crossover_f32.txt) puts its crossover at 13-15 thousand 128-tile tasks
    // (N <= 8192, the measured range of the fp32 single launch): 2 / 5 of the fp64 figure
    if (h->dataflow_max_slots < 0 && h->dtype == 32 && h->Nt <= 64) {
        const long t128 = (long)(h->Nt + 1) * (h->Nt + 2) / 2 * nslots;
        return nslots == 1 || t128 <= (long)h->dataflow_max_tasks * 2 / 5;
    }
    if (nslots > few_slots(h)) {
        // a few more thetas of a SMALL problem still win (fp64 64-tiles): measured crossover at ~2500 tile tasks
        // (N=512: 16 thetas +42 %, 32 +15 %, 64 -16 %; N=1024: 16 +34 %, 32 -9 %)
        const long t64 = (long)(2 * h->Nt + 1) * (2 * h->Nt + 2) / 2 * nslots;
        if (h->dtype != 64 || nslots > 4 * few_slots(h) || t64 > 2500 || h->Nt > h->dataflow_fine_nt) return false;
    }
    if (h->dtype == 32 && h->Nt > 64) return false;        // fp32 has 128-tiles only: measured range ends at N = 8192
    const long tasks = (long)(2 * h->Nt + 1) * (2 * h->Nt + 2) / 2 * nslots;
    return tasks < (1l << 30);
}

// c0 > 0 (128-tiles only): factor the trailing submatrix that starts at tile column c0 -- the tail of the
// look-ahead schedule, already updated by every earlier panel.  No finalize here.
template <typename T, int TBX, int OCC = 2, int NST = 2, bool BUILD = false>
void launch_dataflow(gphip_ctx* h, int nslots, int c0 = 0, double* part = nullptr, long pstride = 0, int ncols = 0, int nprev = 0,
                     const void* aprev = nullptr, unsigned int* colsig = nullptr) {
    const int nd = (int)(h->Npad / TBX) - c0, R = nd + 1;
    // ncols > 0: only the first ncols tile columns (an outer panel of the sharded schedule) -- a prefix of the column-major task list
    // nprev > 0: .. of which the first nprev are a finished panel read through aprev (no tasks of their own)
    const long task0 = (long)nprev * R - (long)nprev * (nprev - 1) / 2;
    const long tasks = ((ncols > 0 && ncols < R ? (long)ncols * R - (long)ncols * (ncols - 1) / 2 : (long)R * (R + 1) / 2) - task0) * nslots;
    DfArgs<T> g{};
    g.nprev = nprev; g.task0 = task0; g.Aprev = (const T*)aprev;
    g.colsig = (TBX == 64 && ncols > 0 && ncols < R) ? colsig : nullptr;
    g.A = (T*)(h->ws_override ? h->ws_override : h->dA); g.bstride = h->slot_elems; g.R128 = (int)h->R; g.c0 = c0;
    g.ncols = (ncols > 0 && ncols < R) ? ncols : 0;
    g.W = (T*)h->dW + (long)c0 * TBX * TBX; g.w_bstride = (long)h->Nt * TB * TB;
    if constexpr (TBX == 64) {
        // the whole factor in one launch for a caller that substitutes afterwards: the 64-block inverses get a buffer of their
        // own (the 128-blocks rebuilt from L go to dW), so that the inverse / forward launches of this kernel find them later
        if (h->want_w && nslots == 1 && c0 == 0 && g.ncols == 0 && nprev == 0) {
            if (!h->dW64 && hipMalloc(&h->dW64, (size_t)h->Nt * TB * TB * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); h->dW64 = nullptr; }
            if (h->dW64) { g.W = (T*)h->dW64; h->w64_gen = h->ws_gen; }
        }
    }
    if constexpr (TBX == 64) {
        g.D = (T*)h->dDinv + (long)c0 * 1024; g.d_bstride = (long)2 * h->Nt * 1024;      // chain hand-over by blocked substitution
    }
    g.partial = h->dPartial + c0; g.p_bstride = h->Npad / TBX;
    if (part) { g.partial = part; g.p_bstride = pstride; }       // (a 64-tile tail keeps its own list of blocks)
    g.info = h->dInfo; g.slotp = h->dSlotp;
    g.flags = h->dFlags; g.f_bstride = (long)(2 * h->Nt + 1) * (2 * h->Nt + 1);
    g.ticket = h->dTicket; g.ticket_base = h->ticket_base;
    g.abort_flag = reinterpret_cast<int*>(h->dTicket + 1);
    g.nd = nd; g.nslots = nslots; g.epoch = ++h->epoch;
    h->ticket_base += (unsigned long long)tasks;
    // profile class 3 (panel work): class 4 stays the trailing SYRK alone, it is what the bench's roofline reads
    ProfScope ps(h, 3, ((double)nd * TBX) * ((double)nd * TBX) * ((double)nd * TBX) / 3.0 * nslots, 0.0);
    ThetaPack tp;
    if constexpr (BUILD) {
        memcpy(tp.v, h->hInvEll, (size_t)nslots * h->d * 8);
        memcpy(tp.v + (size_t)nslots * h->d, h->hSlotp, (size_t)nslots * SLOTP * 8);
        g.xt = (const T*)h->dXt; g.yv = (const T*)h->dY;
        g.n = (int)h->N; g.npad = (int)h->Npad; g.d = (int)h->d; g.kt = h->kt;
        g.hres = h->hRes; g.hinfo = h->hInfo;
        h->hInfo[nslots] = 0;                  // the abort word: only ever SET by the kernel
    }
    size_t lds = df_lds_bytes<T, TBX, NST>();
    if (TBX == 64 && OCC <= 2 && lds < DF_XXF_LDS) lds = DF_XXF_LDS;      // (potrf image behind the stage area, see DF_XXF_POTRF_AT)
    g.park = nullptr;
    if constexpr (TBX == 64) {
        // ONE workgroup per CU while the launch is chain bound: the chain's latency-bound potrf / solve waves then never share
        // a SIMD with another workgroup's back-to-back MFMAs (per-phase stamps, scripts/micro/df_phases.hip: every phase of
        // potrf64 runs 1.6-1.8x slower next to a co-resident accumulating workgroup).  Measured crossover ~3 500 tasks:
        // one theta N = 2048-5120 -3..-7 %, N = 6144 -1 %, N >= 7168 +14 % (throughput bound: two per CU);
        // 2 thetas up to N = 3072, 4 up to 2048, 8 up to 1536.  Occupancy is set through the LDS request (> 80 KiB).
        // (an owner's panel launch of the sharded schedule, chip to itself, has many rows per chain hop: throughput bound down to
        //  far fewer tasks -- owner chain 26.2 -> 22.6 ms at N=32768; under the one-GPU schedule's trailing update no difference)
        // (round 5, after the chain got 30 % shorter: the crossover moved down -- N = 3584 one per CU 0.85 vs 0.87 ms, N = 4096 1.05 vs 1.00)
        const long one_wg_tasks = (g.ncols > 0 && h->dist_world > 0) ? h->df_panel_one_wg_tasks : 1900;
        const int kib = h->dataflow_lds_kib < 0 ? (tasks <= one_wg_tasks ? 84 : 0) : h->dataflow_lds_kib;   // (round 4, after the fence changes: N=4096 1/CU 1.34 vs 1.36, N=5120 1.83 vs 1.72 two per CU)
        if ((size_t)kib * 1024 > lds) lds = (size_t)kib * 1024;
        // two workgroups per CU: the neighbour of a diagonal task steps aside while that task is on the chain
        if (lds <= 80 * 1024 && h->dataflow_park) g.park = reinterpret_cast<int*>(h->dTicket + 2);
    }
    hipLaunchKernelGGL((chol_dataflow_kernel<T, TBX, OCC, NST, BUILD>), dim3((unsigned)tasks), dim3(256), lds, h->stream, g, tp);
}

// U = L^-T of the factor a single dataflow launch has just queued (one slot), by a second launch of the same kernel whose
// tasks are the tiles of U (DfArgs::U): column-major Npad x Npad in dKinv (queue_grad_potri then contracts K^-1 = U U^T into dV).
// The multi-kernel route to the same U (queue_forward_rows over the identity) is a chain of ~3 launches per tile column.
// Column-major U / K^-1 of the inverse-launch route use a leading dimension that is NOT a multiple of a large power of two
// (Npad + 16 elements): with ld = Npad = 8192 doubles every k-column of an operand tile starts 64 KiB after the previous one
constexpr int64_t GRAD_LD_PAD = 16;
// fwd_rows > 0: the FORWARD launch instead -- dV holds fwd_rows right-hand sides as rows (leading dimension fwd_rows), every
// task turns one tile of them into the same tile of V L^-T (DfArgs::u_rows): the forward substitution of a prediction with
// few test points as ONE launch whose chain is a handful of microseconds per 64 columns, not two launches per tile column.
// nslots > 1 (forward only): the same rows against the factors of slots 0 .. nslots-1, slot s's block of right-hand sides at
// dV + s fwd_rows Npad and its 64-block inverses at w64s + s Nt 128^2 (gphip_predict_samples).
template <typename T, int TBX, int OCC = 2, int NST = 2>
void launch_dataflow_inverse(gphip_ctx* h, int64_t fwd_rows = 0, bool back = false, int nslots = 1, const void* w64s = nullptr) {
    const int nd = (int)(h->Npad / TBX);
    const long tasks = (fwd_rows > 0 ? (long)(fwd_rows / TBX) * nd : (long)nd * (nd + 1) / 2) * nslots;
    if (fwd_rows == 0)
        (void)hipMemsetAsync(h->dKinv, 0, (size_t)(h->Npad + GRAD_LD_PAD) * h->Npad * sizeof(T), h->stream);      // (dV stays free for the alpha solve)
    DfArgs<T> g{};
    g.A = (T*)h->dA; g.bstride = h->slot_elems; g.R128 = (int)h->R; g.c0 = 0;
    g.W = (T*)((TBX == 64 && h->dW64 && h->w64_gen == h->ws_gen) ? h->dW64 : h->dW); g.w_bstride = (long)h->Nt * TB * TB;
    g.partial = h->dPartial; g.p_bstride = h->Npad / TBX;
    g.info = h->dInfo; g.slotp = h->dSlotp;
    g.flags = h->dFlags; g.f_bstride = (long)(2 * h->Nt + 1) * (2 * h->Nt + 1);
    g.ticket = h->dTicket; g.ticket_base = h->ticket_base;
    g.abort_flag = reinterpret_cast<int*>(h->dTicket + 1);
    g.nd = nd; g.nslots = nslots; g.epoch = ++h->epoch;
    g.U = (T*)h->dKinv; g.ldu = (long)(h->Npad + GRAD_LD_PAD);
    if (fwd_rows > 0) { g.U = (T*)h->dV; g.ldu = (long)fwd_rows; g.u_rows = (int)(fwd_rows / TBX); g.u_bstride = (long)fwd_rows * h->Npad; }
    if (w64s) g.W = (T*)w64s;
    if (back) { g.u_back = 1; g.LT = (const T*)h->dLT; g.W = (T*)h->dW64T; }        // (the caller made them for this factor: df_backward_ready)
    h->ticket_base += (unsigned long long)tasks;
    ProfScope ps(h, 2, fwd_rows > 0 ? (double)fwd_rows * h->Npad * h->Npad : ((double)h->Npad * h->Npad * h->Npad) / 3.0, 0.0);
    size_t lds = df_lds_bytes<T, TBX, NST>();
    if (TBX == 64 && OCC <= 2 && lds < DF_XXF_LDS) lds = DF_XXF_LDS;
    ThetaPack tp;
    hipLaunchKernelGGL((chol_dataflow_kernel<T, TBX, OCC, NST, false>), dim3((unsigned)tasks), dim3(256), lds, h->stream, g, tp);
    if (fwd_rows == 0) h->u_ready = true;
}

// a prediction of few test points against a factor that came from the 64-tile single launch (its 64-block inverses are still
// there): the forward substitution as one dataflow launch.  Measured against the multi-kernel substitution
// (scripts/gpu_predict_df_sweep.py, profiles/r05_predict_df_sweep.txt): 100 test points 3.1x faster at N = 2048-12288, 2048
// points 1.3-2x, 4096 points 1.1-1.3x up to N = 8192 and a tie at 12288, 8192 points 15-20 % slower
// The backward half of a solve as a dataflow launch needs the factor's 64 x 64 blocks transposed (DfArgs::LT) and the transposed
// 64-block inverses: made once per fit, on the first solve (one pass over the factor, N^2 / 2 elements).  false: no memory.
template <typename T>
bool df_backward_ready(gphip_ctx* h) {
    if (h->dLT && h->dW64T && h->lt_gen == h->ws_gen) return true;
    if (!h->dLT && hipMalloc(&h->dLT, (size_t)h->slot_elems * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); h->dLT = nullptr; return false; }
    if (!h->dW64T && hipMalloc(&h->dW64T, (size_t)h->Nt * TB * TB * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); h->dW64T = nullptr; return false; }
    const long ntiles = (long)h->R * (h->R + 1) / 2;
    hipLaunchKernelGGL(transpose_blocks64_kernel<T>, dim3((unsigned)ntiles), dim3(256), 0, h->stream, (const T*)h->dA, (T*)h->dLT, (long)TS, (int)TB, 4);
    hipLaunchKernelGGL(transpose_blocks64_kernel<T>, dim3((unsigned)(2 * h->Nt)), dim3(256), 0, h->stream, (const T*)h->dW64, (T*)h->dW64T, 4096l, 64, 1);
    h->lt_gen = h->ws_gen;
    return true;
}

// After a fit that did not come from the 64-tile single launch (N > 12288: the look-ahead schedule leaves 128-block inverses in
// dW): cut the 64-block inverses out of them, so that a prediction of few test points is ONE forward dataflow launch there too
// instead of two launches per tile column (N = 16384, 100 test points: 256 launches, 9.1 ms per call).
void ensure_w64(gphip_ctx* h) {
    if (!h->dataflow || h->predict_df <= 0 || h->dtype != 64 || h->dist_world != 0 || h->dist_fit || !has_fit(h) || !h->dW) return;
    if (h->dW64 && h->w64_gen == h->ws_gen) return;
    if (h->Nt > h->predict_df_max_nt) return;
    if (!h->dW64 && hipMalloc(&h->dW64, (size_t)h->Nt * TB * TB * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); h->dW64 = nullptr; return; }
    hipLaunchKernelGGL(w128_to_w64_kernel<double>, dim3((unsigned)(2 * h->Nt)), dim3(256), 0, h->stream, (const double*)h->dW, (double*)h->dW64);
    h->w64_gen = h->ws_gen;
}

bool df_forward_ok(const gphip_ctx* h, int64_t mpad) {
    return h->dataflow && h->predict_df > 0 && h->dtype == 64 && h->dist_world == 0 && h->dW64 && h->w64_gen == h->ws_gen &&
           h->w64_gen == h->fit_gen && mpad <= (h->Npad <= 8192 ? 2 : 1) * (int64_t)h->predict_df && mpad / 64 <= h->Npad / 64;
}

// gphip_predict_samples: nb freshly factored slots (their 128-block inverses are in dW), mpad padded test points each.  true: the
// 64-block inverses of all slots have been cut out of dW into dW64s and ONE forward dataflow launch may serve all slots.
// (fp64, sizes the 64-tile kernel covers; the launch is chain bound per slot, so it pays while it is small: up to 2 048 padded
// rows per slot and 12 000 tasks in all -- beyond that the batched GEMM substitution, whose launches are shared by all slots.
// 8 samples x 100 points: N = 1024 0.75 -> 0.51 ms, N = 2048 1.64 -> 1.11 ms, N = 4096 6.4 -> 5.1 ms per call; 32 x 100: -12 .. -22 %.)
bool samples_forward_df(gphip_ctx* h, int nb, int64_t mpad) {
    if (!h->dataflow || h->predict_df <= 0 || h->dtype != 64 || h->dist_world != 0 || !h->dW || h->Nt > h->dataflow_fine_nt) return false;
    const long nd = h->Npad / 64, tasks = (long)nb * (mpad / 64) * nd;
    if (mpad > h->predict_df || mpad / 64 > nd || tasks > 12000) return false;      // (measured: wins up to ~8 000 tasks, loses from ~16 000)
    const size_t need = (size_t)nb * h->Nt * TB * TB * sizeof(double);
    if (need > h->w64s_bytes) {
        (void)hipFree(h->dW64s);
        h->dW64s = nullptr; h->w64s_bytes = 0;
        if (hipMalloc(&h->dW64s, need) != hipSuccess) { (void)hipGetLastError(); h->dW64s = nullptr; return false; }
        h->w64s_bytes = need;
    }
    hipLaunchKernelGGL(w128_to_w64_kernel<double>, dim3((unsigned)(2 * h->Nt), (unsigned)nb), dim3(256), 0, h->stream, (const double*)h->dW,
                       (double*)h->dW64s, (long)h->Nt * TB * TB);
    return true;
}

// The forward / backward / inverse launches above have no finalize kernel behind them to export the abort word (a dependency
// wait that hit its spin limit makes every later wait of the launch fall through: the results are void).  The caller queues
// this copy behind them and asks for the verdict after its stream synchronisation.
int queue_abort_probe(gphip_ctx* h) {
    HIPCHK(hipMemcpyAsync(h->hInfo + h->slots + 1, reinterpret_cast<int*>(h->dTicket + 1), 4, hipMemcpyDeviceToHost, h->stream));
    return GPHIP_OK;
}
int abort_probe_verdict(gphip_ctx* h, const char* what) {
    if (h->hInfo[h->slots + 1] == 0) return GPHIP_OK;
    h->hInfo[h->slots + 1] = 0;
    HIPCHK(hipMemsetAsync(h->dTicket + 1, 0, 8 + DF_PARK_SLOTS * 4, h->stream));
    h->fitted = false;                         // the factor itself may be intact, but nothing derived from it in this call is
    h->u_ready = false;
    return fail(h, GPHIP_ERR_HIP, what);
}

template <typename T>
void launch_finalize(gphip_ctx* h, int nslots, int nparts, int pstride = 0, const double* part2 = nullptr, int n2 = 0) {
    hipLaunchKernelGGL(finalize_kernel<T>, dim3(nslots), dim3(64), 0, h->stream, (const T*)h->dA, (long)h->slot_elems,
                       (long)(h->slot_elems - TS), h->dPartial, nparts, h->dRes, (const int*)h->dInfo,
                       (const int*)reinterpret_cast<int*>(h->dTicket + 1), h->hRes, h->hInfo, pstride, part2, n2);
}

template <typename T>
int queue_factor_dataflow(gphip_ctx* h, int nslots) {
    h->cs = h->stream;
    if constexpr (sizeof(T) == 8) {
        if (h->Nt <= h->dataflow_fine_nt) {
            // Throughput-bound launches (>= ~8 000 tile tasks: one theta from N = 8192 on) run the build of the same kernel
            // that fits THREE workgroups on a CU (166 registers, a few spills): a resident task holds its slot through the
            // 2-3 column hops it waits for at the end of its life, so more residents = more slots doing work
            // (N=10240 8.79 -> 8.33 ms, N=12288 14.8 -> 13.85 ms, N=8192 -1.8 %; at N=6144, chain bound, +6 %: not there)
            const long t64 = (long)(2 * h->Nt + 1) * (2 * h->Nt + 2) / 2 * nslots;
            const bool occ3 = h->dataflow_occ3 > 0 || (h->dataflow_occ3 < 0 && t64 >= 6000);      // (round 4 re-tune: N=7168 3.07 vs 3.14 ms, N=6144 a tie)
            if (h->fused_eval) {               // tiles built in-kernel, results exported by the corner task
                if (occ3) launch_dataflow<T, 64, 3, 2, true>(h, nslots);
                else launch_dataflow<T, 64, 2, 2, true>(h, nslots);
                return 0;
            }
            if (occ3) launch_dataflow<T, 64, 3>(h, nslots);
            else launch_dataflow<T, 64>(h, nslots);
            launch_finalize<T>(h, nslots, 2 * (int)h->Nt);
            if (h->want_u && nslots == 1) {    // (before the 128-block inverses below overwrite the 64-block ones)
                if (occ3) launch_dataflow_inverse<T, 64, 3>(h);
                else launch_dataflow_inverse<T, 64>(h);
            }
            if (h->want_w)
                hipLaunchKernelGGL(trtri128_kernel<T>, dim3((unsigned)h->Nt, nslots), dim3(256), potrf_lds<T>(), h->stream,
                                   (const T*)h->dA, (long)h->slot_elems, (T*)h->dW, (int)h->Nt);
            return 0;
        }
    }
    // fp64 128-tiles: the out-of-line potrf body needs 232 + 40 registers, so only the 512-register
    // (1 workgroup per CU) build is real -- a 256-register build falls to occupancy 1 anyway AND spills.
    // fp32 accumulators are half the size: 2 workgroups per CU while the schedule is chain bound.
    if constexpr (sizeof(T) == 4) {
        if (h->Nt <= 32) {
            launch_dataflow<T, 128, 2>(h, nslots);
            launch_finalize<T>(h, nslots, (int)h->Nt);
            if (h->want_u && nslots == 1) launch_dataflow_inverse<T, 128, 2>(h);
            return 0;
        }
    }
    if (h->Nt <= 48) launch_dataflow<T, 128, 1, 4>(h, nslots);             // chain bound: deep DMA pipeline
    else launch_dataflow<T, 128, 1>(h, nslots);
    launch_finalize<T>(h, nslots, (int)h->Nt);
    if (h->want_u && nslots == 1) launch_dataflow_inverse<T, 128, 1>(h);
    return 0;
}

// Two-level right-looking Cholesky of slots [0, nslots) (workspace already built on h->stream).
// With look-ahead (default) the panel stream factors panel k+1 while the main stream is still
// applying panel k to everything right of panel k+1:
//   panel stream:  [wait REST(k-1)]  LA(k) = update of panel k+1's columns by panel k;  factor panel k+1
//   main  stream:  [wait panel k]    REST(k) = update of the columns right of panel k+1 by panel k
template <typename T>
int queue_factor(gphip_ctx* h, int nslots) {
    const int Nt = (int)h->Nt, R = Nt + 1;     // R = tile rows incl. the rhs block-row
    const std::vector<int> bnd = panel_bounds(h);
    const int nouter = (int)bnd.size() - 1;
    auto k0 = [&](int k) { return bnd[(size_t)std::min(k, nouter)]; };
    auto trailing = [&](int k, int c_lo, int c_hi, int cls) {      // apply panel k to tile columns [c_lo,c_hi)
        launch_gemm<T>(h, cls, tl<T>(h), tl<T>(h, k0(k)), tl<T>(h, k0(k)), (k0(k + 1) - k0(k)) * TB, c_lo, R, c_lo, c_hi, 1,
                       nslots, 0, 0, Nt);
    };
    if (use_dataflow(h, nslots)) return queue_factor_dataflow<T>(h, nslots);
    double* tail_part = nullptr;               // a 64-tile dataflow tail keeps its block partials here
    int tail_n = 0, tail_k0 = -1;
    // Option "panel_df" (one theta, fp64): every outer panel -- the look-ahead update by the panel before it AND its own
    // factorisation -- is ONE 64-tile dataflow launch on the panel stream (the sharded schedule's dist_panel_df = 2 form).
    // All log-det partials are then per 64-block (dPartial[0 .. 2 Nt), the tail's too).
    bool pdf = false;
    if constexpr (sizeof(T) == 8) pdf = panel_df_on(h, nslots) && nouter >= 2;
    auto df_panel = [&](int kp, int kprev) {
        if constexpr (sizeof(T) == 8) {
            hipStream_t keep = h->stream;
            h->stream = h->pstream;
            const int c0 = 2 * (kprev >= 0 ? k0(kprev) : k0(kp));
            launch_dataflow<T, 64>(h, 1, c0, nullptr, 0, 2 * k0(kp + 1) - c0, kprev >= 0 ? 2 * (k0(kp) - k0(kprev)) : 0, h->dA);
            h->stream = keep;
        }
    };
    if (!h->lookahead || nouter < 2) {
        h->cs = h->stream;
        for (int k = 0; k < nouter; ++k) {
            queue_panel<T>(h, k0(k), k0(k + 1) - k0(k), nslots);
            trailing(k, k0(k + 1), R, 4);
        }
    } else {
        h->sync_used = 0;
        hipEvent_t built = sync_event(h);
        HIPCHK(hipEventRecord(built, h->stream));
        HIPCHK(hipStreamWaitEvent(h->pstream, built, 0));
        h->cs = h->pstream;
        if (pdf) df_panel(0, -1);
        else queue_panel<T>(h, 0, k0(1), nslots);
        hipEvent_t ev_panel = sync_event(h);
        HIPCHK(hipEventRecord(ev_panel, h->pstream));
        hipEvent_t ev_rest = nullptr;
        // tail: once only `dataflow_tail` tile columns are left the dataflow kernel finishes the job in one
        // launch -- the last panels are chain bound, the regime the dataflow schedule wins
        int kc = nouter;
        // (round-4 re-tune: just above the single-launch range a 48-column tail wins -- N=14336: 18.9 vs 19.9 ms with 64 --, from
        //  N=16384 on 64 does: 26.0 vs 26.8)
        const int tail_cols = h->dataflow_tail != 64 ? h->dataflow_tail : (pdf ? 80 : (Nt < 124 ? 48 : 64));
        if (h->dataflow && h->dataflow_tail > 0 && nslots <= few_slots(h) && h->dist_world == 0)
            for (int k = 1; k < nouter; ++k)
                if (Nt - k0(k) <= tail_cols && Nt - k0(k) <= h->dataflow_max_nt) { kc = k; break; }
        for (int k = 0; k < nouter; ++k) {
            hipEvent_t ev_next = nullptr;
            if (k + 1 == kc) {                  // last multi-kernel panel: apply it to everything, then cut over
                h->cs = h->stream;
                HIPCHK(hipStreamWaitEvent(h->stream, ev_panel, 0));
                trailing(k, k0(k + 1), R, 4);
                const int rem = Nt - k0(kc);                       // tile columns left
                if (rem == 0) break;                               // (no tail: this was the last panel -- batches, or dataflow_tail off)
                if constexpr (sizeof(T) == 8) {
                    if (2 * rem <= Nt || pdf) {                    // 64-tiles: the faster chain; its 2 rem block partials
                        tail_part = h->dPartial + (long)h->slots * Nt;         // live behind the 128-block list
                        tail_n = 2 * rem;
                        if (pdf) { tail_part = nullptr; tail_n = 0; }          // (panel_df: one list of 64-blocks for everything)
                        if (h->dataflow_occ3 > 0 || (h->dataflow_occ3 < 0 && (long)(2 * rem + 1) * (2 * rem + 2) / 2 * nslots >= 6000))
                            launch_dataflow<T, 64, 3>(h, nslots, 2 * k0(kc), tail_part, tail_n);
                        else launch_dataflow<T, 64>(h, nslots, 2 * k0(kc), tail_part, tail_n);
                        tail_k0 = k0(kc);
                        break;
                    }
                }
                launch_dataflow<T, 128, 1>(h, nslots, k0(kc));
                break;
            }
            if (k + 1 < nouter) {
                h->cs = h->pstream;
                if (ev_rest) HIPCHK(hipStreamWaitEvent(h->pstream, ev_rest, 0));
                if (pdf) {
                    df_panel(k + 1, k);                                        // LA(k) + factor panel k+1, one launch
                } else {
                    if (h->fuse_potrf && nslots <= 8 && h->dist_world == 0) h->fuse_b = k0(k + 1);  // ... whose first diagonal tile LA(k) also factors
                    trailing(k, k0(k + 1), k0(k + 2), 3);                          // LA(k)
                    const bool first_factored = h->fuse_done;
                    queue_panel<T>(h, k0(k + 1), k0(k + 2) - k0(k + 1), nslots, first_factored);    // factor panel k+1
                }
                ev_next = sync_event(h);
                HIPCHK(hipEventRecord(ev_next, h->pstream));
            }
            h->cs = h->stream;
            HIPCHK(hipStreamWaitEvent(h->stream, ev_panel, 0));
            trailing(k, k0(k + 2), R, 4);                                      // REST(k)
            ev_rest = sync_event(h);
            HIPCHK(hipEventRecord(ev_rest, h->stream));
            ev_panel = ev_next;
        }
        h->cs = h->stream;
    }
    if (pdf) {
        launch_finalize<T>(h, nslots, 2 * Nt);
        if (h->want_w)                             // 64-block inverses everywhere: rebuild the 128-blocks
            hipLaunchKernelGGL(trtri128_kernel<T>, dim3((unsigned)Nt, nslots), dim3(256), potrf_lds<T>(), h->stream,
                               (const T*)h->dA, (long)h->slot_elems, (T*)h->dW, Nt);
    } else if (tail_k0 >= 0) {
        launch_finalize<T>(h, nslots, tail_k0, Nt, tail_part, tail_n);
        if (h->want_w)                             // the tail left 64-block inverses over part of dW: rebuild the 128-blocks
            hipLaunchKernelGGL(trtri128_kernel<T>, dim3((unsigned)Nt, nslots), dim3(256), potrf_lds<T>(), h->stream,
                               (const T*)h->dA, (long)h->slot_elems, (T*)h->dW, Nt);
    } else {
        launch_finalize<T>(h, nslots, Nt);
    }
    return 0;
}

// the staged hyper-parameters of slots [0, nb) -> device (the launches that follow read them there)
int copy_theta(gphip_ctx* h, int nb) {
    HIPCHK(hipMemcpyAsync(h->dInvEll, h->hInvEll, (size_t)nb * h->d * 8, hipMemcpyHostToDevice, h->stream));
    if (h->nl2 > 0) HIPCHK(hipMemcpyAsync(h->dInvEll2, h->hInvEll2, (size_t)nb * h->d * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->dSlotp, h->hSlotp, (size_t)nb * SLOTP * 8, hipMemcpyHostToDevice, h->stream));
    if (h->custom && h->ncp > 0)
        HIPCHK(hipMemcpyAsync(h->dCustomP, h->hCustomP, (size_t)nb * h->ncp * 8, hipMemcpyHostToDevice, h->stream));
    return GPHIP_OK;
}

// stage theta of one slot into the pinned host buffers; returns false if theta is unusable
bool stage_theta(gphip_ctx* h, int slot, const double* th, const double* nug_row = nullptr, const double* mean_row = nullptr) {
    double* ie = h->hInvEll + (size_t)slot * h->d;
    double* ie2 = h->nl2 > 0 ? h->hInvEll2 + (size_t)slot * h->d : nullptr;
    double* sp = h->hSlotp + (size_t)slot * SLOTP;
    bool ok = true;
    for (int i = 0; i < h->p; ++i)
        if (!std::isfinite(th[i])) ok = false;
    if (h->custom) {
        // theta = [p_0 .. p_{ncp-1}] sn [mu]: the function's own parameters go to the device as they are (the points are
        // NOT rescaled: inverse length scales of one); k(x, x), which scales the pivot tolerance, is evaluated on the
        // device (custom_prep_kernel) -- it finds the relative tolerance in sp[3] and the nugget's scale in sp[SP_SF2B]
        for (int j = 0; j < h->d; ++j) ie[j] = 1.0;
        double* cp = h->hCustomP + (size_t)slot * std::max(h->ncp, 1);
        for (int k = 0; k < h->ncp; ++k) cp[k] = ok ? th[k] : 1.0;
        double sn = ok ? th[h->ncp] : 1.0;
        const double mu = (ok && h->mean_id == GPHIP_MEAN_CONST) ? th[h->ncp + 1] : 0.0;
        double nug_scale = sn * sn;
        if (nug_row) {
            nug_scale = 0.0;
            for (int64_t i = 0; i < h->N; ++i) {
                if (!std::isfinite(nug_row[i])) ok = false;
                nug_scale = std::max(nug_scale, std::fabs(nug_row[i]));
            }
        }
        if (mean_row)
            for (int64_t i = 0; i < h->N; ++i)
                if (!std::isfinite(mean_row[i])) ok = false;
        if (!ok || !std::isfinite(nug_scale)) { ok = false; nug_scale = 1.0; sn = 1.0; }
        for (int k = 0; k < SLOTP; ++k) sp[k] = 0.0;
        sp[0] = 1.0; sp[1] = sn * sn; sp[2] = mu; sp[3] = pivot_tol_rel(h); sp[SP_SF2B] = nug_scale; sp[SP_KXX] = 1.0;
        sp[4] = ok ? 0.0 : 1.0;
        return ok;                                 // (sp[SP_MFMA] = 0: cleared above)
    }
    // theta = [term 1: l.., (alpha), sf] [term 2: l.., (alpha), sf] [c] sn [mu]; only |l| matters (l enters squared)
    int o = 0;
    auto lengths = [&](int nl, double* dst) {
        for (int j = 0; j < h->d; ++j) {
            const double l = th[o + (nl == 1 ? 0 : j)];
            if (!(std::fabs(l) > 0.0) || !std::isfinite(1.0 / l)) ok = false;
            dst[j] = ok ? 1.0 / std::fabs(l) : 1.0;
        }
        o += nl;
    };
    double a1 = 1.0, a2 = 1.0, sf2 = 0.0, c = 0.0;
    lengths(h->nl, ie);
    if (h->has_a1) { a1 = th[o++]; if (!(a1 > 0.0)) ok = false; }
    double sf = th[o++];
    if (h->nl2 > 0) {
        lengths(h->nl2, ie2);
        if (h->has_a2) { a2 = th[o++]; if (!(a2 > 0.0)) ok = false; }
        sf2 = th[o++];
    }
    if (h->ks.offset) c = th[o++];
    double sn = th[o++];
    double mu = (h->mean_id == GPHIP_MEAN_CONST) ? th[o++] : 0.0;
    if (!ok) { sf = 1.0; sf2 = 1.0; sn = 1.0; mu = 0.0; a1 = a2 = 1.0; c = 0.0; }
    const double k1 = sf * sf, k2 = sf2 * sf2;
    const double kxx = (h->ks.op == 1 ? k1 + k2 : (h->ks.op == 2 ? k1 * k2 : k1)) + c;      // k(x, x): every family is 1 at r = 0
    sp[0] = k1; sp[1] = sn * sn; sp[2] = mu;
    sp[SP_SF2B] = k2; sp[SP_ALPHA1] = a1; sp[SP_ALPHA2] = a2; sp[SP_OFFSET] = c; sp[SP_KXX] = kxx;
    double nug_scale = sn * sn, nug_min = sn * sn;
    if (nug_row) {                             // point-dependent nugget: the pivot tolerance scales with its largest value
        nug_scale = 0.0;
        nug_min = HUGE_VAL;
        for (int64_t i = 0; i < h->N; ++i) {
            if (!std::isfinite(nug_row[i])) ok = false;
            nug_scale = std::max(nug_scale, std::fabs(nug_row[i]));
            nug_min = std::min(nug_min, std::fabs(nug_row[i]));
        }
    }
    if (mean_row)
        for (int64_t i = 0; i < h->N; ++i)
            if (!std::isfinite(mean_row[i])) ok = false;
    if (!ok) nug_scale = 1.0;
    sp[3] = pivot_tol_rel(h) * (std::fabs(kxx) + nug_scale);
    if (!std::isfinite(kxx) || !std::isfinite(sp[1]) || !std::isfinite(k1) || !std::isfinite(k2)) {
        ok = false;
        sp[0] = sp[1] = sp[SP_KXX] = 1.0; sp[SP_SF2B] = 0.0; sp[SP_OFFSET] = 0.0; sp[3] = 1e-14;
    }
    sp[4] = ok ? 0.0 : 1.0;
    // which kernel builds this slot's K (see gphip_ctx::kbuild_mfma)
    sp[SP_MFMA] = 0.0;
    if (h->kbuild_mfma && mfma_family(h) >= 0 && h->d <= KB_LDS_MAXD && h->dCentre && ok) {
        double bound = 0.0;
        for (int j = 0; j < h->d; ++j) bound += (h->x_half[(size_t)j] * ie[j]) * (h->x_half[(size_t)j] * ie[j]);
        const double lim = h->dtype == 64 ? (double)h->kbuild_mfma_bound : (double)h->kbuild_mfma_bound / 8.0;
        // entry accuracy (bound <= lim) AND what the conditioning of K makes of it (see gphip_ctx::kbuild_mfma_digits);
        // fp32 results carry eps32 cond(K) whichever kernel builds K: the same rule 6 digits up
        const double eps = h->dtype == 64 ? 2.220446049250313e-16 : 5.9604644775390625e-8;
        const double amp = eps * std::max(bound, 64.0) * (1.0 + std::fabs(kxx) / nug_min);      // (nug_min = 0: inf, the direct form)
        const double amp_lim = std::pow(10.0, -(double)h->kbuild_mfma_digits + (h->dtype == 64 ? 0.0 : 6.0));
        if (h->kbuild_mfma >= 2 || (bound <= lim && amp <= amp_lim)) sp[SP_MFMA] = 1.0;
    }
    return ok;
}

template <typename T>
int queue_null_reduce(gphip_ctx* h, int B) {
    hipLaunchKernelGGL(null_reduce_kernel<T>, dim3(B), dim3(1024), 0, h->stream, (const T*)h->dY, (int)h->N, h->dNullMu,
                       h->dNullOut);
    return 0;
}

template <typename T>
int upload_rows(gphip_ctx* h, void* dst, const double* src, int nb, int64_t n, int64_t npad) {
    std::vector<T> tmp((size_t)nb * npad, (T)0);
    for (int s = 0; s < nb; ++s)
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)s * npad + i] = (T)src[(size_t)s * n + i];
    HIPCHK(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GPHIP_OK;
}

// device copies of the point-dependent nugget / mean rows [s0, s0 + nb) of the current call (one row per workspace slot)
int upload_pw(gphip_ctx* h, int s0, int nb) {
    h->pw_mean_on = h->pw_nug_on = false;
    if (!h->pw_mean_host && !h->pw_nug_host) return GPHIP_OK;
    const int cap = std::max(h->slots, nb);
    if (h->pw_cap < cap) {
        (void)hipFree(h->dPwMean); (void)hipFree(h->dPwNug);
        h->dPwMean = h->dPwNug = nullptr; h->pw_cap = 0;
        HIPCHK(hipMalloc(&h->dPwMean, (size_t)cap * h->Npad * h->es));
        HIPCHK(hipMalloc(&h->dPwNug, (size_t)cap * h->Npad * h->es));
        h->pw_cap = cap;
    }
    int rc = GPHIP_OK;
    if (h->pw_mean_host) {
        rc = DISPATCH(h, upload_rows, h, h->dPwMean, h->pw_mean_host + (size_t)s0 * h->N, nb, h->N, h->Npad);
        if (rc) return rc;
        h->pw_mean_on = true;
    }
    if (h->pw_nug_host) {
        rc = DISPATCH(h, upload_rows, h, h->dPwNug, h->pw_nug_host + (size_t)s0 * h->N, nb, h->N, h->Npad);
        if (rc) return rc;
        h->pw_nug_on = true;
    }
    return GPHIP_OK;
}

template <typename T>
int queue_null_reduce_pw(gphip_ctx* h, int B, const double* c_nug, const double* nug, const double* mean, double* out) {
    hipLaunchKernelGGL(null_reduce_pw_kernel<T>, dim3(B), dim3(1024), 0, h->stream, (const T*)h->dY, (int)h->N, h->dNullMu, c_nug,
                       nug, mean, out);
    return 0;
}

// null kernel with a point-dependent nugget and / or mean: K = diag(nu_i) (BGP:25-27, 156-159 with nugget /@ points)
int null_kernel_batch_pw(gphip_ctx* h, const double* Theta, int B, double* out, double* parts, int* info) {
    const int64_t N = h->N;
    double *dNug = nullptr, *dMean = nullptr, *dC = nullptr, *dMu = nullptr, *dOut = nullptr;
    auto cleanup = [&]() { (void)hipFree(dNug); (void)hipFree(dMean); (void)hipFree(dC); (void)hipFree(dMu); (void)hipFree(dOut); };
    std::vector<double> cn((size_t)B), mu((size_t)B), sums((size_t)3 * B);
    for (int s = 0; s < B; ++s) {
        const double sn = Theta[(size_t)s * h->p];
        cn[(size_t)s] = sn * sn;
        const double m = (h->mean_id == GPHIP_MEAN_CONST) ? Theta[(size_t)s * h->p + 1] : 0.0;
        mu[(size_t)s] = std::isfinite(m) ? m : 0.0;
    }
    int rc = GPHIP_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == GPHIP_OK) { h->err = hipGetErrorString(e); rc = GPHIP_ERR_HIP; } };
    chk(hipMalloc(&dC, (size_t)B * 8)); chk(hipMalloc(&dMu, (size_t)B * 8)); chk(hipMalloc(&dOut, (size_t)B * 24));
    if (h->pw_nug_host) chk(hipMalloc(&dNug, (size_t)B * N * 8));
    if (h->pw_mean_host) chk(hipMalloc(&dMean, (size_t)B * N * 8));
    if (rc == GPHIP_OK) {
        chk(hipMemcpyAsync(dC, cn.data(), (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
        chk(hipMemcpyAsync(dMu, mu.data(), (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
        if (dNug) chk(hipMemcpyAsync(dNug, h->pw_nug_host, (size_t)B * N * 8, hipMemcpyHostToDevice, h->stream));
        if (dMean) chk(hipMemcpyAsync(dMean, h->pw_mean_host, (size_t)B * N * 8, hipMemcpyHostToDevice, h->stream));
    }
    if (rc == GPHIP_OK) {
        double* keep = h->dNullMu;
        h->dNullMu = dMu;
        DISPATCH(h, queue_null_reduce_pw, h, B, dC, dNug, dMean, dOut);
        h->dNullMu = keep;
        chk(hipMemcpyAsync(sums.data(), dOut, (size_t)B * 24, hipMemcpyDeviceToHost, h->stream));
        chk(hipStreamSynchronize(h->stream));
        chk(hipGetLastError());
    }
    cleanup();
    if (rc) return rc;
    for (int s = 0; s < B; ++s) {
        const double* th = Theta + (size_t)s * h->p;
        const double logdet = sums[(size_t)3 * s], quad = sums[(size_t)3 * s + 1];
        const double ll = -0.5 * ((double)N * LOG_TWO_PI + logdet + quad);
        bool finite_theta = true;
        for (int i = 0; i < h->p; ++i) finite_theta = finite_theta && std::isfinite(th[i]);
        info[s] = (finite_theta && std::isfinite(ll)) ? (sums[(size_t)3 * s + 2] == 0.0 ? GPHIP_INFO_OK : GPHIP_INFO_NOT_SPD) : GPHIP_INFO_NAN;
        out[s] = ll;
        if (parts) { parts[2 * s] = logdet; parts[2 * s + 1] = quad; }
    }
    return GPHIP_OK;
}

// null kernel Function[0] (BGP:25-27, 156-159): K = diag(sn^2).  The residual sums are reduced on the
// device from the resident y (one workgroup per theta); the scalar epilogue is the usual host side of
// the ABI.  grad (optional, B x p): d/dsn = (quad - N)/sn, d/dmu = sum(y - mu)/sn^2.
int null_kernel_batch(gphip_ctx* h, const double* Theta, int B, double* out, double* parts, int* info, double* grad) {
    if ((h->pw_mean_host || h->pw_nug_host) && !grad) return null_kernel_batch_pw(h, Theta, B, out, parts, info);
    if (B > h->null_cap) {
        (void)hipFree(h->dNullMu); (void)hipFree(h->dNullOut);
        h->dNullMu = h->dNullOut = nullptr; h->null_cap = 0;
        HIPCHK(hipMalloc(&h->dNullMu, (size_t)B * 8));
        HIPCHK(hipMalloc(&h->dNullOut, (size_t)B * 16));
        h->null_cap = B;
    }
    std::vector<double> mu((size_t)B), sums((size_t)2 * B);
    for (int s = 0; s < B; ++s) {
        const double m = (h->mean_id == GPHIP_MEAN_CONST) ? Theta[(size_t)s * h->p + 1] : 0.0;
        mu[(size_t)s] = std::isfinite(m) ? m : 0.0;
    }
    HIPCHK(hipMemcpyAsync(h->dNullMu, mu.data(), (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
    DISPATCH(h, queue_null_reduce, h, B);
    HIPCHK(hipMemcpyAsync(sums.data(), h->dNullOut, (size_t)B * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    for (int s = 0; s < B; ++s) {
        const double* th = Theta + (size_t)s * h->p;
        const double sn = th[0], v = sn * sn;
        const double logdet = (double)h->N * std::log(std::fabs(v));
        const double quad = sums[(size_t)2 * s + 1] / v;
        const double ll = -0.5 * ((double)h->N * LOG_TWO_PI + logdet + quad);
        bool finite_theta = true;
        for (int i = 0; i < h->p; ++i) finite_theta = finite_theta && std::isfinite(th[i]);
        info[s] = (finite_theta && std::isfinite(ll)) ? (v > 0.0 ? GPHIP_INFO_OK : GPHIP_INFO_NOT_SPD) : GPHIP_INFO_NAN;
        out[s] = ll;
        if (parts) { parts[2 * s] = logdet; parts[2 * s + 1] = quad; }
        if (grad) {
            grad[(size_t)s * h->p] = (quad - (double)h->N) / sn;
            if (h->mean_id == GPHIP_MEAN_CONST) grad[(size_t)s * h->p + 1] = sums[(size_t)2 * s] / v;
        }
    }
    return GPHIP_OK;
}

// s0: index of the chunk's first theta within the call (rows of the call's point-dependent nugget / mean arrays)
int eval_chunk(gphip_ctx* h, const double* Theta, int nb, double* out, double* parts, int* info, int s0 = 0) {
    std::vector<char> okv(nb);
    for (int s = 0; s < nb; ++s)
        okv[s] = stage_theta(h, s, Theta + (size_t)s * h->p, h->pw_nug_host ? h->pw_nug_host + (size_t)(s0 + s) * h->N : nullptr,
                             h->pw_mean_host ? h->pw_mean_host + (size_t)(s0 + s) * h->N : nullptr);
    {
        const int rc = upload_pw(h, s0, nb);
        if (rc) return rc;
    }
    // few thetas: they travel as kernel arguments of the first kernel (no copies, no memset); the results
    // come back through pinned host memory written by the finalize kernel (no copies either)
    h->theta_packed = h->kt != 2 && (size_t)nb * (h->d + SLOTP) <= (size_t)THETA_PACK;
    if (!h->theta_packed) {
        const int rc = copy_theta(h, nb);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(h->dInfo, 0, (size_t)nb * 4, h->stream));
    }
    // one theta (or a few), fp64, small enough for 64-tiles and nobody needs the scaled inputs / 128-block
    // inverses afterwards: the evaluation is ONE launch
    h->fused_eval = h->fuse_option && h->theta_packed && h->kt != 2 && h->dtype == 64 && !h->want_w && h->profile < 2 &&
                    !h->pw_mean_on && !h->pw_nug_on && use_dataflow(h, nb) && h->Nt <= h->dataflow_fine_nt;
    h->cs = h->stream;
    {
        ProfScope ps(h, 5, 0.0, 0.0);
        if (!h->fused_eval) DISPATCH(h, queue_build, h, nb);
        DISPATCH(h, queue_factor, h, nb);
    }
    h->fused_eval = false;
    h->theta_packed = false;
    h->pw_mean_on = h->pw_nug_on = false;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    harvest(h);
    if (h->hInfo[nb] != 0 && getenv("GPHIP_DEBUG")) {
        fprintf(stderr, "gphip debug: nb=%d hInfo[nb]=%d slots=%d; hInfo[0..]:", nb, h->hInfo[nb], h->slots);
        for (int i = 0; i <= nb && i < 80; ++i) fprintf(stderr, " %d", h->hInfo[i]);
        fprintf(stderr, "\n");
    }
    if (h->hInfo[nb] != 0) {                   // a dataflow dependency wait hit its spin limit: results are void
        HIPCHK(hipMemsetAsync(h->dTicket + 1, 0, 8 + DF_PARK_SLOTS * 4, h->stream));
        return fail(h, GPHIP_ERR_HIP, "dataflow Cholesky schedule timed out (set option dataflow=0 and report)");
    }
    for (int s = 0; s < nb; ++s) {
        const double logdet = h->hRes[2 * s], quad = h->hRes[2 * s + 1];
        const double ll = -0.5 * ((double)h->N * LOG_TWO_PI + logdet + quad);
        int inf = h->hInfo[s];
        if (!okv[s]) inf = GPHIP_INFO_NAN;
        else if (inf == 0 && !std::isfinite(ll)) inf = GPHIP_INFO_NAN;
        info[s] = inf;
        out[s] = ll;
        if (parts) { parts[2 * s] = logdet; parts[2 * s + 1] = quad; }
    }
    return GPHIP_OK;
}

int eval_batch_local(gphip_ctx* h, const double* Theta, int B, int p, double* out, double* parts, int* info) {
    if (!h || !Theta || !out || !info) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length for this kernel/mean");
    if (B <= 0) return GPHIP_OK;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    if (h->kernel_id == GPHIP_KERNEL_NULL) return null_kernel_batch(h, Theta, B, out, parts, info, nullptr);
    int rc = ensure_slots(h, B);
    if (rc) return rc;
    invalidate_fit(h);
    for (int s0 = 0; s0 < B; s0 += h->slots) {
        const int nb = (B - s0 < h->slots) ? (B - s0) : h->slots;
        rc = eval_chunk(h, Theta + (size_t)s0 * p, nb, out + s0, parts ? parts + 2 * s0 : nullptr, info + s0, s0);
        if (rc) return rc;
    }
    return GPHIP_OK;
}

template <typename T>
int set_func_attrs(gphip_ctx* h) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf128_kernel<T>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)potrf_lds<T>()));
    constexpr int df128 = (int)df_lds_bytes<T, 128, 2>(), df128x = (int)df_lds_bytes<T, 128, 4>();
    if constexpr (sizeof(T) == 4)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_dataflow_kernel<T, 128, 2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, df128));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_dataflow_kernel<T, 128, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, df128));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>((chol_dataflow_kernel<T, 128, 1, 4>)),
                               hipFuncAttributeMaxDynamicSharedMemorySize, df128x));
    if constexpr (sizeof(T) == 8) {
        constexpr int df64 = 152 * 1024;       // (room for the one-workgroup-per-CU request, option "dataflow_lds_kib")
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_dataflow_kernel<T, 64, 2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, df64));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_dataflow_kernel<T, 64, 3>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, df64));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>((chol_dataflow_kernel<T, 64, 3, 2, true>)),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, df64));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>((chol_dataflow_kernel<T, 64, 2, 2, true>)),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, df64));
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trtri128_kernel<T>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)potrf_lds<T>()));
    if (h->kt == 2 && h->d <= KB_LDS_MAXD) {   // general covariance form: both terms' row and column points in LDS
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>((kbuild_kernel<T, 0, 2>)),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * h->d * TB * sizeof(T))));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(grad_reduce_general_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((4 * h->d + 1) * TB * sizeof(T) + 8 + 4 * (2 * h->d + 6) * 8)));     // (+ GradEmit's staging area)
    }
#define GEMM_ATTR(ROLE)                                                                                   \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, ROLE, 2, 2, 2>),             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS));                  \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, ROLE, 4, 4, 4>),             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * GEMM_LDS)));
    GEMM_ATTR(0) GEMM_ATTR(1) GEMM_ATTR(2) GEMM_ATTR(3)
#undef GEMM_ATTR
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, 4, 2, 2, 2>),     // (+ the fused potrf image)
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(GEMM_LDS, potrf_lds<T>())));
    return GPHIP_OK;
}

// host fp64 -> device T upload / download helpers (synchronous on `st`)
template <typename T>
int upload(gphip_ctx* h, void* dst, const std::vector<double>& src, hipStream_t st) {
    if (sizeof(T) == 8) {
        HIPCHK(hipMemcpyAsync(dst, src.data(), src.size() * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    } else {
        std::vector<float> tmp(src.begin(), src.end());
        HIPCHK(hipMemcpyAsync(dst, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return GPHIP_OK;
}

template <typename T>
int download(gphip_ctx* h, std::vector<double>& dst, const void* src, size_t n, hipStream_t st) {
    dst.resize(n);
    if (sizeof(T) == 8) {
        HIPCHK(hipMemcpyAsync(dst.data(), src, n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else {
        std::vector<float> tmp(n);
        HIPCHK(hipMemcpyAsync(tmp.data(), src, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (size_t i = 0; i < n; ++i) dst[i] = tmp[i];
    }
    return GPHIP_OK;
}

int ensure_vbuf(gphip_ctx* h, int64_t cap) {
    if (cap <= h->vcap) return GPHIP_OK;
    (void)hipFree(h->dV); (void)hipFree(h->dXsT); (void)hipFree(h->dXsS); (void)hipFree(h->dMean); (void)hipFree(h->dVar);
    (void)hipFree(h->dPwMeanT); (void)hipFree(h->dPwNugT); (void)hipFree(h->dXsS2);
    h->dXsS2 = nullptr;
    h->dV = h->dXsT = h->dXsS = nullptr;
    h->dMean = h->dVar = h->dPwMeanT = h->dPwNugT = nullptr;
    h->vcap = 0;
    HIPCHK(hipMalloc(&h->dV, (size_t)cap * h->Npad * h->es));
    HIPCHK(hipMalloc(&h->dXsT, (size_t)cap * h->d * h->es));
    HIPCHK(hipMalloc(&h->dXsS, (size_t)cap * h->d * h->es));
    if (h->nl2 > 0) HIPCHK(hipMalloc(&h->dXsS2, (size_t)cap * h->d * h->es));
    HIPCHK(hipMalloc(&h->dMean, (size_t)cap * 8));
    HIPCHK(hipMalloc(&h->dVar, (size_t)cap * 8));
    HIPCHK(hipMalloc(&h->dPwMeanT, (size_t)cap * 8));
    HIPCHK(hipMalloc(&h->dPwNugT, (size_t)cap * 8));
    h->vcap = cap;
    return GPHIP_OK;
}

// One outer panel [k0, k1) of the forward substitution below: solve its tile columns, update the rest of the panel after
// each, then update everything right of the panel ONCE with K = 128 (k1 - k0).  L is read through tl<T>() -- the dense
// workspace, or (sharded prediction) whatever buffer currently holds this panel (ws_override).
template <typename T>
int queue_forward_panel(gphip_ctx* h, int64_t mpad, int nslots, int k0, int k1, int b_start, bool identity_rows) {
    const int Nt = (int)h->Nt, Mt = (int)(mpad / TB);
    const long vs = (long)mpad * h->Npad, lrs = (long)Nt * TB * TB;
    T *V = (T*)h->dV, *W = (T*)h->dW;
    auto rows_at = [&](int b) { return identity_rows ? std::min(Mt, b - b_start + 1) : Mt; };
    for (int b = k0; b < k1; ++b) {
        launch_gemm<T>(h, 2, cm<T>(V, mpad, vs), cm<T>(V + (long)b * TB * mpad, mpad, vs), wb<T>(W, b, lrs), TB, 0, rows_at(b), b,
                       b + 1, 0, nslots, 1);
        if (b + 1 < k1)
            launch_gemm<T>(h, 3, cm<T>(V, mpad, vs), cm<T>(V + (long)b * TB * mpad, mpad, vs), tl<T>(h, b), TB, 0, rows_at(b),
                           b + 1, k1, 0, nslots);
    }
    if (k1 < Nt)
        launch_gemm<T>(h, 3, cm<T>(V, mpad, vs), cm<T>(V + (long)k0 * TB * mpad, mpad, vs), tl<T>(h, k0), (k1 - k0) * TB, 0,
                       rows_at(k1 - 1), k1, Nt, 0, nslots);
    return 0;
}

// V <- V L^-T for the mpad x Npad row block in dV (right-looking over the 128-tile columns of L):
// every row of V becomes (L^-1 v)^T.  Panel solves and updates are the same MFMA GEMM kernel.
template <typename T>
int queue_forward_rows(gphip_ctx* h, int64_t mpad, int nslots, int b_start = 0, bool identity_rows = false) {
    // Two-level like the factorisation: inside an outer panel of `panel` tile columns the updates have
    // K = 128 and touch the panel only; everything right of the panel is updated ONCE per panel with
    // K = 128*panel.  (Single-level, every block column re-read and re-wrote all of V to its right:
    // HBM bound -- 1.3 TB of traffic for N = 65536, M = 10000 in fp32.)
    // identity_rows: V holds rows b_start*128.. of the identity, so the result (rows of L^-T) is upper
    // triangular -- at tile column b only the row tiles <= b - b_start are non-zero and are touched.
    // many rows (prediction of thousands of test points, K^-1 for the gradient): every pass over the columns right of a
    // panel reads and writes all of V there, so wider panels pay (cfg 5, M = 10 000: 374 -> 359 ms from 4 to 12 tiles)
    const int Nt = (int)h->Nt, Mt = (int)(mpad / TB), P = (Mt >= 8 && h->panel_wide) ? std::max(h->panel, 12) : h->panel;
    for (int k0 = b_start; k0 < Nt; k0 += P)     // b_start > 0: the rows are known to be zero left of tile column b_start
        queue_forward_panel<T>(h, mpad, nslots, k0, std::min(k0 + P, Nt), b_start, identity_rows);
    return 0;
}

// V <- V L^-1 (backward substitution, block columns from last to first), "NN" GEMM role:
//   X_b = Y_b W_b ;  Y_c -= X_b L(b,c) for c < b.   Two-level as above.
template <typename T>
int queue_backward_rows(gphip_ctx* h, int64_t mpad) {
    const int Nt = (int)h->Nt, Mt = (int)(mpad / TB), P = (Mt >= 8 && h->panel_wide) ? std::max(h->panel, 12) : h->panel;
    T *V = (T*)h->dV, *W = (T*)h->dW;
    for (int k1 = Nt; k1 > 0; k1 -= P) {         // outer panel = tile columns [k0, k1)
        const int k0 = (k1 - P > 0) ? k1 - P : 0;
        for (int b = k1 - 1; b >= k0; --b) {
            launch_gemm<T>(h, 6, cm<T>(V, mpad, 0), cm<T>(V + (long)b * TB * mpad, mpad, 0), cm<T>(W, TB, 0), TB, 0, Mt, b, b + 1, 0,
                           1, 1);
            if (b > k0)                         // (the J operand is L(b, c) read transposed: tile (b, c) of the workspace)
                launch_gemm<T>(h, 6, cm<T>(V, mpad, 0), cm<T>(V + (long)b * TB * mpad, mpad, 0), tl<T>(h, b, false), TB, 0, Mt, k0, b,
                               0, 1, 0);
        }
        if (k0 > 0)                             // Y_c -= X[:, k0..k1) L(k0..k1, c) for every c < k0
            launch_gemm<T>(h, 6, cm<T>(V, mpad, 0), cm<T>(V + (long)k0 * TB * mpad, mpad, 0), tl<T>(h, k0, false), (k1 - k0) * TB, 0,
                           Mt, 0, k0, 0, 1, 0);
    }
    return 0;
}

// slot 0's scaled training inputs xs = x / l (what queue_build leaves behind), without building K
template <typename T>
int queue_scale_train(gphip_ctx* h) {
    const long tot = (long)h->d * h->Npad;
    int gx = (int)((tot + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_scale<T>, dim3(gx, 1), dim3(256), 0, h->cs, (const T*)h->dXt, (T*)h->dXs, h->dInvEll, (int)h->d,
                       (int)h->Npad);
    if (h->nl2 > 0)
        hipLaunchKernelGGL(k_scale<T>, dim3(gx, 1), dim3(256), 0, h->cs, (const T*)h->dXt, (T*)h->dXs2, h->dInvEll2, (int)h->d,
                           (int)h->Npad);
    return 0;
}

// the test points about to be uploaded ([d][mpad], mc of them): how far outside the training inputs' range do they lie?
void note_test_range(gphip_ctx* h, const std::vector<double>& xt, int64_t mc, int64_t mpad) {
    double r = 0.0;
    if (h->dCentre)
        for (int64_t j = 0; j < h->d; ++j) {
            const double c = h->x_centre[(size_t)j], half = h->x_half[(size_t)j];
            for (int64_t i = 0; i < mc; ++i) {
                const double dev = std::fabs(xt[(size_t)j * mpad + i] - c);
                if (dev > 0.0) r = std::max(r, half > 0.0 ? dev / half : HUGE_VAL);
            }
        }
    h->test_ratio = r;
}

// V(t, j) = k_theta_s(x*_t, x_j) for every slot s: the (unscaled) test points in dXsT are scaled by
// each slot's 1/l into dXsS[slot]
template <typename T>
int queue_cross(gphip_ctx* h, int64_t mc, int64_t mpad, int nslots) {
    const long tot = (long)h->d * mpad;
    hipLaunchKernelGGL(k_scale<T>, dim3((unsigned)((tot + 255) / 256), nslots), dim3(256), 0, h->stream,
                       (const T*)h->dXsT, (T*)h->dXsS, h->dInvEll, (int)h->d, (int)mpad);
    if (h->nl2 > 0)
        hipLaunchKernelGGL(k_scale<T>, dim3((unsigned)((tot + 255) / 256), nslots), dim3(256), 0, h->stream,
                           (const T*)h->dXsT, (T*)h->dXsS2, h->dInvEll2, (int)h->d, (int)mpad);
    KBuildArgs<T> a{};
    a.ks = h->ks; a.xi2 = (const T*)h->dXsS2; a.xj2 = (const T*)h->dXs2;
    a.out = (T*)h->dV; a.ld = mpad; a.bstride = (long)mpad * h->Npad;
    a.xi = (const T*)h->dXsS; a.xj = (const T*)h->dXs; a.xi_bstride = tot; a.xj_bstride = (long)h->d * h->Npad;
    a.npad_i = (int)mpad; a.npad_j = (int)h->Npad; a.n_i = (int)mc; a.n_j = (int)h->N;
    a.y = nullptr; a.slotp = h->dSlotp; a.d = (int)h->d; a.mode = 1; a.nt_i = (int)(mpad / TB); a.nt_j = (int)h->Nt;
    a.exp2tab = h->dExp2;
    // (test points far outside the training inputs' range would void the bound the slots' SP_MFMA verdicts rest on)
    const bool far = !(h->test_ratio <= 2.0);
    launch_kbuild<T>(h, a, dim3((unsigned)((mpad / TB) * h->Nt), nslots), far ? nullptr : (const T*)h->dXsT, (const T*)h->dXt);
    return 0;
}

// m(x*) / nugget(x*) of the test points [m0, m0 + mc) of the samples [s0, s0 + nb) -> device [nb][mpad] (pw_*_test rows are
// pw_test_stride apart)
int upload_pw_test(gphip_ctx* h, int s0, int nb, int64_t m0, int64_t mc, int64_t mpad) {
    for (int which = 0; which < 2; ++which) {
        const double* src = which ? h->pw_nug_test : h->pw_mean_test;
        if (!src) continue;
        std::vector<double> tmp((size_t)nb * mpad, 0.0);
        for (int s = 0; s < nb; ++s)
            for (int64_t t = 0; t < mc; ++t) tmp[(size_t)s * mpad + t] = src[(size_t)(s0 + s) * h->pw_test_stride + m0 + t];
        HIPCHK(hipMemcpyAsync(which ? h->dPwNugT : h->dPwMeanT, tmp.data(), tmp.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GPHIP_OK;
}

// run-time compiled covariance function: k(x*_t, x*_t) of the test points in dXsT for every slot -> dKss[slot][mpad] (fp64)
int queue_custom_kss(gphip_ctx* h, int64_t mc, int64_t mpad, int nslots) {
    const size_t want = (size_t)nslots * mpad * 8;
    if (want > h->kss_cap) {
        (void)hipFree(h->dKss);
        h->dKss = nullptr; h->kss_cap = 0;
        HIPCHK(hipMalloc(&h->dKss, want));
        h->kss_cap = want;
    }
    const void* x = h->dXsT;                       // the (unscaled) test points [d][mpad], shared by the slots
    long xbs = 0, ostride = (long)mpad;
    int npad = (int)mpad, n = (int)mc, d = (int)h->d, ncp = std::max(h->ncp, 1);
    const double* cp = h->dCustomP;
    double* out = h->dKss;
    void* params[] = {&x, &xbs, &npad, &n, &d, &cp, &ncp, &out, &ostride};
    HIPCHK(hipModuleLaunchKernel(h->f_cdiag, (unsigned)((mc + 255) / 256), (unsigned)nslots, 1, 256, 1, 1, 0, h->stream, params, nullptr));
    return GPHIP_OK;
}

// mu*, var* from V and z: V streamed once (HBM bound) -- strips of columns x 128 test points per workgroup,
// then the strips are added in order.  Profile class 6: bytes = V once.
template <typename T>
int queue_predict_reduce(gphip_ctx* h, int64_t mc, int64_t mpad, int nslots) {
    const int Mt = (int)(mpad / TB), Nt = (int)h->Nt;
    int nstrips = (2048 + Mt * nslots - 1) / (Mt * nslots);          // enough workgroups to fill 256 CUs several times
    if (nstrips > Nt) nstrips = Nt;
    if (nstrips > 64) nstrips = 64;
    if (nstrips < 1) nstrips = 1;
    int js = (Nt + nstrips - 1) / nstrips * TB;
    if (js > 4096) js = 4096;
    nstrips = (int)((h->Npad + js - 1) / js);
    const size_t need = (size_t)nslots * nstrips * 2 * mpad * 8;
    if (need > h->part_cap) {
        (void)hipFree(h->dPart);
        h->dPart = nullptr; h->part_cap = 0;
        HIPCHK(hipMalloc(&h->dPart, need));
        h->part_cap = need;
    }
    const double* kss = nullptr;
    if (h->custom) {                               // k(x*, x*) is a function of the test point for a general covariance function
        const int rc = queue_custom_kss(h, mc, mpad, nslots);
        if (rc) return rc;
        kss = h->dKss;
    }
    {
        ProfScope ps(h, 6, 4.0 * (double)mpad * h->Npad * nslots, (double)sizeof(T) * mpad * h->Npad * nslots);
        hipLaunchKernelGGL(predict_partial_kernel<T>, dim3((unsigned)Mt, (unsigned)nstrips, (unsigned)nslots), dim3(256),
                           (size_t)js * 8 + 8 * TB * 8, h->stream, (const T*)h->dV, (long)mpad, (long)mpad * h->Npad, (int)h->N,
                           h->z_vector ? (const T*)h->dZ : (const T*)h->dA, h->z_vector ? 0 : (int)h->R, (long)h->slot_elems, js,
                           h->dPart, nstrips);
        hipLaunchKernelGGL(predict_finish_kernel, dim3((unsigned)((mc + 255) / 256), (unsigned)nslots), dim3(256), 0, h->stream,
                           (const double*)h->dPart, nstrips, (long)mpad, (const double*)h->dSlotp, (int)mc, (long)mpad,
                           h->dMean, h->dVar, h->pw_mean_test ? (const double*)h->dPwMeanT : nullptr,
                           h->pw_nug_test ? (const double*)h->dPwNugT : nullptr, kss);
    }
    return 0;
}

template <typename T>
int queue_dist_update(gphip_ctx* h, const void* packed, long K0, long rows, long cols, int c_lo, int c_hi, int cls,
                      int groups = 1, int grp_stride = 0) {
    // the packed panel is the panel's own contiguous range of the tile-major workspace: shifted base, global tile indices
    (void)rows;
    Opnd<T> pk{(const T*)packed - tile_index((int)K0, (int)K0, (int)h->R) * TS, TB, 0, (int)h->R, (int)K0};
    // C = this rank's own panels: the dense workspace, or its compact own-panel storage addressed through the adj table
    Opnd<T> co{(const T*)h->dist_base, TB, 0, (int)h->R, 0};
    co.adj = h->dDistAdj; co.adj_panel = h->panel;
    launch_gemm<T>(h, cls, co, pk, pk, (int)cols, c_lo, (int)h->Nt + 1, c_lo, c_hi, 1, 1, 0, 0, (int)h->Nt, groups,
                   grp_stride, h->panel);
    return 0;
}

template <typename T>
int queue_finalize(gphip_ctx* h) {         // sharded evaluation: the corner tile lives in rank 0's storage (slot nouter)
    const int nouter = (int)((h->Nt + h->panel - 1) / h->panel);
    const long corner = h->dist_rank == 0 ? (h->dist_adj[(size_t)nouter] + tile_index((int)h->Nt, (int)h->Nt, (int)h->R)) * TS : 0l;
    // (dataflow panels: log-det partials per 64-block, 2 Nt entries; un-owned entries stay zero either way)
    hipLaunchKernelGGL(finalize_kernel<T>, dim3(1), dim3(64), 0, h->stream, (const T*)h->dist_base, 0l, corner, h->dPartial,
                       (int)(h->dist_df_active ? 2 * h->Nt : h->Nt), h->dRes);
    return 0;
}

template <typename T, int KT>
void launch_grad_kt(gphip_ctx* h, const GradArgs<T>& a, dim3 grid, size_t lds) {
#define GR_CASE(DD)                                                                                   \
    case DD:                                                                                          \
        hipLaunchKernelGGL((grad_reduce_kernel<T, DD, KT>), grid, dim3(256), lds, h->cs, a);          \
        break;
    switch (a.d) {
        GR_CASE(1) GR_CASE(2) GR_CASE(3) GR_CASE(4) GR_CASE(5) GR_CASE(6) GR_CASE(7) GR_CASE(8) GR_CASE(16)
        default:
            hipLaunchKernelGGL((grad_reduce_kernel<T, 0, KT>), grid, dim3(256), lds, h->cs, a);
    }
#undef GR_CASE
}

// where the accumulators of a gradient reduction go (GradEmit, gp_kernels.h): per-workgroup rows in dGpart + a [4][np] staging
// area behind the kernel's own LDS -- or, when that area would be larger than 8 KiB or the rows cannot be allocated, the atomics
// on gacc.  Returns the LDS bytes to launch with.
template <typename T>
size_t grad_rows(gphip_ctx* h, GradArgs<T>& a, dim3 grid, size_t lds) {
    const size_t np = h->ngacc, nwg = (size_t)grid.x * grid.y, need = nwg * np * 8;
    a.gpart = nullptr; a.np = (int)np; a.ws_off = (int)((lds + 7) / 8);
    if (np == 0 || 4 * np * 8 > 8192) return lds;
    if (need > h->gpart_bytes) {
        (void)hipFree(h->dGpart);
        h->dGpart = nullptr; h->gpart_bytes = 0;
        if (hipMalloc((void**)&h->dGpart, need) != hipSuccess) { (void)hipGetLastError(); h->dGpart = nullptr; return lds; }
        h->gpart_bytes = need;
    }
    a.gpart = h->dGpart;
    return (size_t)a.ws_off * 8 + 4 * np * 8;
}
template <typename T>
void grad_rows_finish(gphip_ctx* h, const GradArgs<T>& a, dim3 grid) {
    if (a.gpart)
        hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)a.np), dim3(256), 0, h->cs, (const double*)a.gpart, (long)grid.x * grid.y, a.np, a.gacc);
}

template <typename T>
void launch_grad(gphip_ctx* h, GradArgs<T>& a, dim3 grid) {
    if (h->custom) {                               // the run-time compiled custom_grad_kernel<T> (dual-number instantiation)
        const double* cp = h->dCustomP;
        int ncp = h->ncp;
        const size_t lds = grad_rows<T>(h, a, grid, (size_t)((a.d > KB_LDS_MAXD ? 0 : a.d) + 1) * TB * sizeof(T));
        void* params[] = {&a, &cp, &ncp};
        (void)hipModuleLaunchKernel(h->f_cgrad, grid.x, grid.y, grid.z, 256, 1, 1, (unsigned)lds, h->cs, params, nullptr);
        grad_rows_finish<T>(h, a, grid);
        return;
    }
    if (a.d > KB_LDS_MAXD) {
        // any form, more dimensions than the specialised kernels hold in LDS / registers: the general kernel reading the points
        // from global memory, one launch per window of 32 length-scale derivatives
        a.ks = h->ks; a.xs2 = (const T*)h->dXs2;
        const size_t lds = grad_rows<T>(h, a, grid, (size_t)TB * sizeof(T));
        for (a.d0 = 0; a.d0 < a.d; a.d0 += 32) {
            hipLaunchKernelGGL(grad_reduce_general_kernel<T>, grid, dim3(256), lds, h->cs, a);
            grad_rows_finish<T>(h, a, grid);
        }
        a.d0 = 0;
        return;
    }
    if (h->kt == 0 || h->kt == 1) {
        const size_t lds = grad_rows<T>(h, a, grid, (size_t)(a.d + 1) * TB * sizeof(T));
        if (h->kt == 0) launch_grad_kt<T, 0>(h, a, grid, lds);
        else launch_grad_kt<T, 1>(h, a, grid, lds);
    } else {
        a.ks = h->ks; a.xs2 = (const T*)h->dXs2;
        const size_t lds = grad_rows<T>(h, a, grid, (size_t)(4 * a.d + 1) * TB * sizeof(T));
        hipLaunchKernelGGL(grad_reduce_general_kernel<T>, grid, dim3(256), lds, h->cs, a);
    }
    grad_rows_finish<T>(h, a, grid);
}

// ---- substitutions with 1 .. TRSV_MAXR right-hand sides: one launch per triangle, L streamed once (gp_trsv.h) ----
// The factor of slot 0 and its 128-block inverses (dW) must be resident.  false: not applicable / no memory (the caller keeps
// the GEMM-shaped substitution).
// Scratch (one allocation, dTrsvX):  [input block: TRSV_MAXR x Npad] [pass 0] [pass 1],  a pass = {X, Xc: nrhs x Npad each,
// S: nrhs x 128 x (Nt (Nt - 1) / 2 + 1), ticket: 64 bytes} laid out for the call's nrhs -- everything a launch polls or counts
// with is ONE contiguous range per call, filled with the sentinel (0xFF bytes) by ONE memset for both passes: the ticket too,
// the kernel counts from 0xFFFFFFFF + 1.
size_t trsv_pass_elems(const gphip_ctx* h, int nrhs) {
    return (size_t)nrhs * ((size_t)2 * h->Npad + (size_t)TB * (size_t)(h->Nt * (h->Nt - 1) / 2 + 1)) + 64 / h->es;
}
bool trsv_ok(gphip_ctx* h, int nrhs) {
    if (!h->trsv || nrhs < 1 || nrhs > TRSV_MAXR || h->Nt > 512 || h->dist_world > 0 || h->ws_override) return false;
    const size_t bytes = ((size_t)TRSV_MAXR * h->Npad + 2 * trsv_pass_elems(h, TRSV_MAXR)) * h->es;
    if (!h->dTrsvX && hipMalloc(&h->dTrsvX, bytes) != hipSuccess) { (void)hipGetLastError(); h->dTrsvX = nullptr; return false; }
    if (!h->dTrsvP && hipMalloc(&h->dTrsvP, (size_t)4 * h->Nt * TS * h->es) != hipSuccess) { (void)hipGetLastError(); h->dTrsvP = nullptr; return false; }
    return true;
}
template <typename T> T* trsv_input(gphip_ctx* h) { return (T*)h->dTrsvX; }
template <typename T> T* trsv_pass(gphip_ctx* h, int nrhs, int pass) { return (T*)h->dTrsvX + (size_t)TRSV_MAXR * h->Npad + (size_t)pass * trsv_pass_elems(h, nrhs); }

// the sentinel for `npasses` launches of a call with nrhs right-hand sides
template <typename T>
int queue_trsv_fill(gphip_ctx* h, int nrhs, int npasses) {
    HIPCHK(hipMemsetAsync(trsv_pass<T>(h, nrhs, 0), 0xFF, (size_t)npasses * trsv_pass_elems(h, nrhs) * sizeof(T), h->stream));
    return GPHIP_OK;
}

// pass `pass` of the call: X <- L^-1 B (back = false) or L^-T B (back = true); B: [nrhs][Npad]; returns X ([nrhs][Npad]) in *Xout
template <typename T>
int queue_trsv(gphip_ctx* h, const T* B, int pass, int nrhs, bool back, T** Xout) {
    const int nt = (int)h->Nt;
    if (h->trsvp_gen != h->ws_gen) {             // first substitution with this factor: the chain's products, both directions
        for (int dir = 0; dir < 2; ++dir)
            hipLaunchKernelGGL(trsv_prep_kernel<T>, dim3((unsigned)nt, 2), dim3(256), 0, h->stream, (const T*)h->dA, (int)h->R, (const T*)h->dW,
                               (T*)h->dTrsvP + (size_t)dir * 2 * nt * TS, nt, dir);
        h->trsvp_gen = h->ws_gen;
    }
    T* base = trsv_pass<T>(h, nrhs, pass);
    TrsvArgs<T> g{};
    g.A = (const T*)h->dA; g.R128 = (int)h->R; g.W = (const T*)h->dW; g.B = B; g.ldx = (long)h->Npad;
    g.P = (const T*)h->dTrsvP + (size_t)(back ? 1 : 0) * 2 * nt * TS;
    g.X = base; g.Xc = base + (size_t)nrhs * h->Npad; g.S = base + (size_t)2 * nrhs * h->Npad;
    g.ticket = reinterpret_cast<unsigned int*>(g.S + (size_t)nrhs * TB * (size_t)(nt * (nt - 1) / 2 + 1));
    g.nt = nt; g.nrhs = nrhs; g.back = back ? 1 : 0; g.dbg = 0;              // (dbg: developer timing bits, scripts/micro/trsv_trace.hip only)
    g.abort_flag = reinterpret_cast<int*>(h->dTicket + 1);
    const long ntasks = nt >= 5 ? (long)(nt - 4) * (nt - 3) / 2 : 0;                 // common ticket list: I >= K + 4
    const long grid = std::min<long>((long)h->ncu, 3 * TRSV_CHAIN + ntasks);         // chain pairs + feeders + tile role; one per CU: all resident
    ProfScope ps(h, 2, 0.0, (double)h->slot_elems * sizeof(T));
    if (back) hipLaunchKernelGGL((trsv_dataflow_kernel<T, true>), dim3((unsigned)grid), dim3(TRSV_THREADS), trsv_lds_bytes(sizeof(T)), h->stream, g);
    else hipLaunchKernelGGL((trsv_dataflow_kernel<T, false>), dim3((unsigned)grid), dim3(TRSV_THREADS), trsv_lds_bytes(sizeof(T)), h->stream, g);
    *Xout = g.X;
    return GPHIP_OK;
}

// alpha = K^-1 r from the fitted factor (z = L^-1 r sits in the rhs row): one backward pass on a
// 128-row scratch block whose row 0 is z
template <typename T>
int queue_alpha(gphip_ctx* h) {
    const int64_t mpad = TB, Npad = h->Npad;
    if (trsv_ok(h, 1)) {                       // one backward launch that streams L once
        T *z = trsv_input<T>(h), *x = nullptr;
        hipLaunchKernelGGL(gather_rhs_row_kernel<T>, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, h->stream, (const T*)h->dA,
                           (int)h->R, 0, (int)Npad, z, 1l);
        int rc = queue_trsv_fill<T>(h, 1, 1);
        if (!rc) rc = queue_trsv<T>(h, z, 0, 1, true, &x);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h->dAlpha, x, (size_t)Npad * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
        return GPHIP_OK;
    }
    HIPCHK(hipMemsetAsync(h->dV, 0, (size_t)mpad * Npad * sizeof(T), h->stream));
    hipLaunchKernelGGL(gather_rhs_row_kernel<T>, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, h->stream, (const T*)h->dA,
                       (int)h->R, 0, (int)Npad, (T*)h->dV, (long)mpad);
    queue_backward_rows<T>(h, mpad);
    HIPCHK(hipMemcpy2DAsync(h->dAlpha, sizeof(T), h->dV, (size_t)mpad * sizeof(T), sizeof(T), (size_t)Npad,
                            hipMemcpyDeviceToDevice, h->stream));
    return GPHIP_OK;
}

// rows [c0, c0+mc) of K^-1 into dV (identity rows -> forward -> backward), then the reduction
template <typename T>
int queue_grad_chunk(gphip_ctx* h, int64_t c0, int64_t mc, int64_t mpad) {
    const long tot = (long)mpad * h->Npad;
    int gx = (int)((tot + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(identity_rows_kernel<T>, dim3(gx), dim3(256), 0, h->stream, (T*)h->dV, (long)mpad, (int)h->Npad,
                       (int)c0, (int)mc);
    queue_forward_rows<T>(h, mpad, 1, (int)(c0 / TB), true);
    queue_backward_rows<T>(h, mpad);
    GradArgs<T> a{};
    a.Kinv = (const T*)h->dV; a.ldv = mpad; a.alpha = (const T*)h->dAlpha; a.xs = (const T*)h->dXs;
    a.npad = (int)h->Npad; a.n = (int)h->N; a.c0 = (int)c0; a.mc = (int)mc; a.d = (int)h->d;
    a.slotp = h->dSlotp; a.gacc = h->dGacc;
    const dim3 grid((unsigned)(mpad / TB), (unsigned)h->Nt);
    launch_grad<T>(h, a, grid);
    return GPHIP_OK;
}

// alpha = L^-T z with the explicit U = L^-T that launch_dataflow_inverse left in dKinv (z = the factored right-hand-side row);
// scratch: the head of dV (z, then the per-chunk partial sums), free until K^-1 is contracted into it
template <typename T>
int queue_alpha_from_u(gphip_ctx* h) {
    const int npad = (int)h->Npad, chunk = 128, nch = (npad + chunk - 1) / chunk;     // (128-column chunks: Nt x Nt / 2 workgroups with work)
    T* z = (T*)h->dV;
    double* part = reinterpret_cast<double*>(static_cast<char*>(h->dV) + (((size_t)npad * sizeof(T) + 255) / 256) * 256);
    hipLaunchKernelGGL(gather_rhs_row_kernel<T>, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, (const T*)h->dA,
                       (int)h->R, 0, npad, z, 1l);
    hipLaunchKernelGGL(utri_gemv_partial_kernel<T>, dim3((unsigned)(npad / TB), (unsigned)nch), dim3(TB), 0, h->stream,
                       (const T*)h->dKinv, (long)(npad + GRAD_LD_PAD), (const T*)z, npad, chunk, part);
    hipLaunchKernelGGL(utri_gemv_finish_kernel<T>, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, (const double*)part, nch,
                       npad, (T*)h->dAlpha);
    return GPHIP_OK;
}

// The whole of K^-1 at once, LAPACK potri style (2/3 N^3 instead of the 4/3 N^3 of forward + backward
// substitution): U = L^-T from a forward pass over all identity rows (upper triangular, zero tiles
// skipped), then the lower tiles of K^-1 = U U^T as ONE triangular launch whose tile (i,j) contracts
// k >= 128 i only, then the reduction over the lower triangle (strictly lower tiles counted twice).
template <typename T>
int queue_grad_potri(gphip_ctx* h) {
    const long npad = h->Npad;
    const long tot = npad * npad;
    int gx = (int)((tot + 255) / 256);
    if (gx > 4096) gx = 4096;
    // a single-launch factorisation has left U = L^-T in dKinv already (launch_dataflow_inverse): K^-1 then goes to dV
    const bool pre = h->u_ready;
    h->u_ready = false;
    const T* Ub = (const T*)(pre ? h->dKinv : h->dV);
    const T* Kb = (const T*)(pre ? h->dV : h->dKinv);
    if (!pre) {
        hipLaunchKernelGGL(identity_rows_kernel<T>, dim3(gx), dim3(256), 0, h->stream, (T*)h->dV, npad, (int)npad, 0, (int)h->N);
        queue_forward_rows<T>(h, npad, 1, 0, true);
    }
    const long ldk = pre ? npad + GRAD_LD_PAD : npad;
    launch_gemm<T>(h, 2, cm<T>(Kb, ldk, 0), cm<T>(Ub, ldk, 0), cm<T>(Ub, ldk, 0),
                   (int)npad, 0, (int)h->Nt, 0, (int)h->Nt, 1, 1, 1, 1);
    GradArgs<T> a{};
    a.Kinv = Kb; a.ldv = ldk; a.alpha = (const T*)h->dAlpha; a.xs = (const T*)h->dXs;
    a.npad = (int)npad; a.n = (int)h->N; a.c0 = 0; a.mc = (int)h->N; a.d = (int)h->d; a.tri = 1;
    a.slotp = h->dSlotp; a.gacc = h->dGacc;
    const dim3 grid((unsigned)h->Nt, (unsigned)h->Nt);
    launch_grad<T>(h, a, grid);
    return GPHIP_OK;
}

void apply_env_options(gphip_ctx* h);        // GPHIP_OPTIONS, defined next to gphip_set_option

// tiles of outer panel k (its contiguous range of the packed layout) and the dense index of its first tile
long dist_panel_tiles(const gphip_ctx* h, int k) {
    const int K0 = k * h->panel, K1 = (int)std::min<int64_t>(K0 + h->panel, h->Nt);
    return tile_index(K1, K1, (int)h->R) - tile_index(K0, K0, (int)h->R);
}
long dist_panel_first(const gphip_ctx* h, int k) { return tile_index(k * h->panel, k * h->panel, (int)h->R); }

// Lay out where this rank keeps its panels for (rank, world, panel width): compact own-panel storage, or the dense
// workspace when the factor is to be replicated.  dist_adj[q] = (local tile offset of slot q) - (its dense tile index).
int dist_layout(gphip_ctx* h, int rank, int world, bool full) {
    const int nouter = (int)((h->Nt + h->panel - 1) / h->panel);
    if (h->lay_rank == rank && h->lay_world == world && h->lay_panel == h->panel && h->lay_full == (int)full &&
        (full ? h->dA != nullptr : h->dOwn != nullptr)) {
        h->dist_base = full ? h->dA : h->dOwn;
        return GPHIP_OK;
    }
    // any failure below leaves NO layout behind (a draining member must never address panel storage through a stale one)
    h->lay_rank = -1; h->lay_world = 0; h->lay_full = -1;
    h->dist_base = nullptr;
    h->dist_adj.assign((size_t)nouter + 1, 0);
    if (full) {
        int rc = ensure_slots(h, 1, true);
        if (rc) return rc;
        (void)hipFree(h->dDistAdj);
        h->dDistAdj = nullptr;
        h->dist_base = h->dA;
    } else {
        long off = 0;
        for (int q = 0; q < nouter; ++q)
            if (q % world == rank) {
                h->dist_adj[(size_t)q] = off - dist_panel_first(h, q);
                off += dist_panel_tiles(h, q);
            }
        if (rank == 0) {                                    // the rhs x rhs corner tile
            h->dist_adj[(size_t)nouter] = off - tile_index((int)h->Nt, (int)h->Nt, (int)h->R);
            off += 1;
        }
        if (off < 1) off = 1;                               // (a rank that owns nothing still gets a valid pointer)
        const size_t need = (size_t)off * TS * h->es;
        if (need > h->own_bytes) {
            (void)hipFree(h->dOwn);
            h->dOwn = nullptr; h->own_bytes = 0;
            HIPCHK(hipMalloc(&h->dOwn, need));
            h->own_bytes = need;
        }
        if (!h->dDistAdj || h->lay_panel != h->panel) {
            (void)hipFree(h->dDistAdj);
            h->dDistAdj = nullptr;
            HIPCHK(hipMalloc(&h->dDistAdj, ((size_t)nouter + 1) * sizeof(long)));
        }
        HIPCHK(hipMemcpyAsync(h->dDistAdj, h->dist_adj.data(), ((size_t)nouter + 1) * sizeof(long), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->dist_base = h->dOwn;
    }
    h->lay_rank = rank; h->lay_world = world; h->lay_panel = h->panel; h->lay_full = (int)full;
    return GPHIP_OK;
}

// base pointer through which the tiles of OWNED outer panel q are addressed with their global (dense) tile indices
char* dist_panel_base(const gphip_ctx* h, int q) {
    return static_cast<char*>(h->dist_base) + h->dist_adj[(size_t)q] * TS * (long)h->es;
}
// the panel's own contiguous range (what is broadcast)
char* dist_panel_range(const gphip_ctx* h, int q) {
    return dist_panel_base(h, q) + dist_panel_first(h, q) * TS * (long)h->es;
}

}  // namespace

// Developer experiment, never set in production: a context whose streams may only use a SLICE of the chip's CUs, so that virtual
// ranks sharing ONE GPU run side by side instead of time-slicing the whole chip -- the closest thing to several GPUs a one-GPU
// box offers (scripts/gpu_cu_partition.py).  A slice is 1 / W of the CUs of EVERY XCD: bit b of a
// hipExtStreamCreateWithCUMask mask is CU b / 8 of XCD b % 8 (profiles/r04_cumask_map.txt), slice r of W = bits
// [256 r / W, 256 (r + 1) / W) -- whole shader-engine rounds, so the dispatcher's even split over engines stays balanced.
// (Masks that empty a whole XCD are ignored by this stack: scripts/micro/cumask.hip.)  GPHIP_CU_SLICE="r/W": every context of
// the process on that slice; GPHIP_CU_PARTITION=1: member i of a W-member one-device group on slice i (set by group_create).
// Both are read only in a process started with GPHIP_TEST_HOOKS=1.
static thread_local int g_slice_r = 0, g_slice_w = 0;
static bool dev_hooks() {                   // like the fault-injection options: only in a process started with GPHIP_TEST_HOOKS=1
    static const bool on = [] { const char* e = getenv("GPHIP_TEST_HOOKS"); return e && !strcmp(e, "1"); }();
    return on;
}
static bool cu_slice(int* r, int* w) {
    if (!dev_hooks()) return false;
    if (g_slice_w) { *r = g_slice_r; *w = g_slice_w; return true; }
    const char* e = getenv("GPHIP_CU_SLICE");
    return e && sscanf(e, "%d/%d", r, w) == 2 && (*w == 2 || *w == 4 || *w == 8) && *r >= 0 && *r < *w;
}
static hipError_t make_slice_stream(hipStream_t* st, int r, int w) {
    uint32_t words[8] = {0};
    for (int b = 256 * r / w; b < 256 * (r + 1) / w; ++b) words[b / 32] |= 1u << (b % 32);
    const hipError_t e = hipExtStreamCreateWithCUMask(st, 8, words);
    static bool told = false;
    if (!told) fprintf(stderr, "gphip: DEVELOPER CU partition active (slice %d of %d on a new stream: %s)\n", r, w, hipGetErrorString(e));
    told = true;
    return e;
}

static int create_ctx(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id, int dtype, int device,
                      gphip_handle* out, const char* custom_body = nullptr, int ncp = 0, std::string* why = nullptr);
namespace {
// the covariance function of the gphip_create_custom* call in progress on this thread: every context the call creates (one per
// local device of a multi-device handle) compiles it for itself
thread_local const char* g_pending_body = nullptr;
thread_local int g_pending_ncp = 0;
thread_local std::string g_create_error;
}
#include "gphip_multi.inc"

namespace {
// A fit made by ONE context of a multi-device handle (below shard_min_n, gradient, local refit) starts a new fit of the
// group: the other members' factors (an earlier sharded fit, possibly of the same theta but another nugget / mean
// array) must not serve its predictions.
void stamp_fit(gphip_ctx* h) {
    h->fit_gen = h->ws_gen;
    if (h->group && !h->in_group_call && !h->group->in_sharded_fit) h->fit_id = ++h->group->fit_id;
}
}  // namespace

namespace {
// every likelihood-type entry point lands here: a plain handle evaluates locally; a group handle shards ONE
// factorisation over its ranks (N >= shard_min_n) or deals a batch of thetas to its local devices
int eval_batch(gphip_ctx* h, const double* Theta, int B, int p, double* out, double* parts, int* info) {
    if (h && h->group && Theta && out && info && B >= 1 && h->kernel_id != GPHIP_KERNEL_NULL) {
        if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length for this kernel/mean");
        if (B == 1 && group_shards(h)) return group_eval(h, Theta, p, h->want_w, out, parts, info);
        if (B > 1 && !h->want_w) return group_eval_batch(h, Theta, B, p, out, parts, info);
    }
    return eval_batch_local(h, Theta, B, p, out, parts, info);
}
}  // namespace

extern "C" {

const char* gphip_version(void) { return "gphip 0.5.0 (gfx950; fp64 + fp32; multi-device; tile-major; MFMA kernel build)"; }

int gphip_device_count(int* n) {
    if (!n) return GPHIP_ERR_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return GPHIP_OK;
}

const char* gphip_last_error(gphip_handle h) { return h ? h->err.c_str() : "null handle"; }

// one plain context on one device: data upload, streams, kernel attributes
static int create_ctx(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id, int dtype,
                      int device /* < 0: current */, gphip_handle* out, const char* custom_body, int ncp, std::string* why) {
    if (!out) return GPHIP_ERR_ARG;
    *out = nullptr;
    if (!X || !y) return GPHIP_ERR_ARG;
    if (!custom_body && kernel_id == GPHIP_KERNEL_CUSTOM && g_pending_body) {
        custom_body = g_pending_body; ncp = g_pending_ncp; why = &g_create_error;
    }
    if (custom_body) kernel_id = GPHIP_KERNEL_CUSTOM;
    else if (kernel_id == GPHIP_KERNEL_CUSTOM) return GPHIP_ERR_ARG;      // (only gphip_create_custom* make such a handle)
    if (N < 1 || d < 1) return GPHIP_ERR_DIM;
    // (d > KB_LDS_MAXD = 32: the kernel build reads the point tiles from global memory instead of LDS; gradients stay limited)
    // kernel_id: a plain named kernel, or GPHIP_KERNEL_COMPOSE(term1, op, term2, offset)
    struct Base { int fam; bool ard; };
    auto base = [](int id, Base& b) {
        switch (id) {
            case GPHIP_KERNEL_SE: b = {0, false}; return true;
            case GPHIP_KERNEL_SE_ARD: b = {0, true}; return true;
            case GPHIP_KERNEL_MATERN52: b = {1, false}; return true;
            case GPHIP_KERNEL_MATERN52_ARD: b = {1, true}; return true;
            case GPHIP_KERNEL_MATERN32: b = {2, false}; return true;
            case GPHIP_KERNEL_MATERN32_ARD: b = {2, true}; return true;
            case GPHIP_KERNEL_RQ: b = {3, false}; return true;
            case GPHIP_KERNEL_RQ_ARD: b = {3, true}; return true;
            default: return false;
        }
    };
    const bool composed = kernel_id >= (1 << 24);
    const int id1 = composed ? (kernel_id & 0xff) : kernel_id, id2 = composed ? ((kernel_id >> 8) & 0xff) : 0;
    const int op = composed ? ((kernel_id >> 16) & 0xf) : 0, offs = composed ? ((kernel_id >> 20) & 0xf) : 0;
    Base b1{0, false}, b2{0, false};
    if (kernel_id != GPHIP_KERNEL_NULL && !custom_body) {
        if (kernel_id < 0 || !base(id1, b1)) return GPHIP_ERR_ARG;
        if (op < 0 || op > 2 || offs > 1 || (op != 0 && !base(id2, b2))) return GPHIP_ERR_ARG;
    }
    if (mean_id != GPHIP_MEAN_ZERO && mean_id != GPHIP_MEAN_CONST) return GPHIP_ERR_ARG;
    if (dtype != 64 && dtype != 32) return GPHIP_ERR_UNSUPPORTED;
    int ndevs = 0;
    if (hipGetDeviceCount(&ndevs) != hipSuccess || ndevs < 1) return GPHIP_ERR_NODEVICE;
    gphip_ctx* h = new gphip_ctx;
    if (device >= 0) h->device = device;
    else (void)hipGetDevice(&h->device);
    if (h->device < 0 || h->device >= ndevs) { delete h; return GPHIP_ERR_NODEVICE; }
    if (hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->ncu < 1) {
        (void)hipGetLastError();
        h->ncu = 256;
    }
    h->dtype = dtype;
    h->es = dtype == 64 ? 8 : 4;
    h->N = N; h->d = d;
    h->Npad = (N + TB - 1) / TB * TB;
    h->Nt = h->Npad / TB;
    h->R = h->Nt + 1;
    h->slot_elems = h->R * (h->R + 1) / 2 * TS;
    h->kernel_id = kernel_id; h->mean_id = mean_id;
    if (kernel_id == GPHIP_KERNEL_NULL) {
        h->kt = 0; h->nl = 0;
        h->p = 1 + (mean_id == GPHIP_MEAN_CONST ? 1 : 0);
    } else if (custom_body) {
        // a covariance function given as source text: the general-form code paths (kt 2) around a kernel build compiled below
        h->custom = true; h->ncp = ncp;
        h->ks = KSpec{4, 0, 0, 0};
        h->kt = 2; h->nl = (int)d; h->nl2 = 0;
        h->p = ncp + 1 + (mean_id == GPHIP_MEAN_CONST ? 1 : 0);
    } else {
        // SE / Matern-5/2 alone keep their specialised kernels (kt 0 / 1); everything else runs the general form (kt 2)
        h->ks = KSpec{b1.fam, b2.fam, op, offs};
        h->kt = (op == 0 && offs == 0 && b1.fam <= 1) ? b1.fam : 2;
        h->nl = b1.ard ? (int)d : 1;
        h->nl2 = op != 0 ? (b2.ard ? (int)d : 1) : 0;
        h->has_a1 = b1.fam == 3;
        h->has_a2 = op != 0 && b2.fam == 3;
        h->p = h->nl + (h->has_a1 ? 1 : 0) + 1 + (op != 0 ? h->nl2 + (h->has_a2 ? 1 : 0) + 1 : 0) + offs + 1 +
               (mean_id == GPHIP_MEAN_CONST ? 1 : 0);
    }
    const double* Xd = static_cast<const double*>(X);
    const double* yd = static_cast<const double*>(y);
    auto bail = [&](int code) { gphip_destroy(h); return code; };
    if (hipSetDevice(h->device) != hipSuccess) return bail(GPHIP_ERR_HIP);
    if (int sr = 0, sw = 0; cu_slice(&sr, &sw)) {
        // developer experiment (scripts/gpu_cu_partition.py): this context runs on a slice of the chip's CUs
        if (make_slice_stream(&h->stream, sr, sw) != hipSuccess || make_slice_stream(&h->pstream, sr, sw) != hipSuccess) return bail(GPHIP_ERR_HIP);
        h->cs = h->stream;
    } else {
        int least = 0, greatest = 0;                // numerically lower = higher priority
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, least) != hipSuccess) return bail(GPHIP_ERR_HIP);
        if (hipStreamCreateWithPriority(&h->pstream, hipStreamNonBlocking, greatest) != hipSuccess) return bail(GPHIP_ERR_HIP);
        h->cs = h->stream;
    }
    std::vector<double> xt((size_t)d * h->Npad, 0.0), yp((size_t)h->Npad, 0.0);
    for (int64_t i = 0; i < N; ++i) {
        for (int64_t j = 0; j < d; ++j) xt[(size_t)j * h->Npad + i] = Xd[i * d + j];
        yp[i] = yd[i];
    }
    if (hipMalloc(&h->dXt, xt.size() * h->es) != hipSuccess) return bail(GPHIP_ERR_HIP);
    if (hipMalloc(&h->dY, yp.size() * h->es) != hipSuccess) return bail(GPHIP_ERR_HIP);
    {
        std::vector<double> tab(EXP_TAB);
        for (int j = 0; j < EXP_TAB; ++j) tab[(size_t)j] = std::exp2((double)j / EXP_TAB);
        if (hipMalloc(&h->dExp2, tab.size() * 8) != hipSuccess) return bail(GPHIP_ERR_HIP);
        if (hipMemcpy(h->dExp2, tab.data(), tab.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(GPHIP_ERR_HIP);
    }
    if (DISPATCH(h, upload, h, h->dXt, xt, h->stream) != GPHIP_OK) return bail(GPHIP_ERR_HIP);
    if (DISPATCH(h, upload, h, h->dY, yp, h->stream) != GPHIP_OK) return bail(GPHIP_ERR_HIP);
    if (kernel_id != GPHIP_KERNEL_NULL && !custom_body && mfma_family(h) >= 0) {
        // mid-range / half range of the inputs AS THE DEVICE HOLDS THEM (fp32 handles: rounded to float)
        h->x_centre.assign((size_t)d, 0.0); h->x_half.assign((size_t)d, 0.0);
        bool finite = true;
        for (int64_t j = 0; j < d; ++j) {
            double lo = 0.0, hi = 0.0;
            for (int64_t i = 0; i < N; ++i) {
                const double v = dtype == 64 ? xt[(size_t)j * h->Npad + i] : (double)(float)xt[(size_t)j * h->Npad + i];
                if (i == 0 || v < lo) lo = v;
                if (i == 0 || v > hi) hi = v;
                if (!std::isfinite(v)) finite = false;
            }
            double c = 0.5 * lo + 0.5 * hi;
            if (dtype == 32) c = (double)(float)c;
            h->x_centre[(size_t)j] = c;
            h->x_half[(size_t)j] = std::max(hi - c, c - lo);
        }
        if (finite) {
            if (hipMalloc(&h->dCentre, (size_t)d * 8) != hipSuccess) return bail(GPHIP_ERR_HIP);
            if (hipMemcpy(h->dCentre, h->x_centre.data(), (size_t)d * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(GPHIP_ERR_HIP);
        }
    }
    if (DISPATCH(h, set_func_attrs, h) != GPHIP_OK) return bail(GPHIP_ERR_HIP);
    if (custom_body) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return bail(GPHIP_ERR_HIP);
        std::string arch = prop.gcnArchName;                   // "gfx950:sramecc+:xnack-" -> "gfx950"
        arch = arch.substr(0, arch.find(':'));
        std::string msg;
        h->custom_body = custom_body; h->arch = arch;
        const std::shared_ptr<const RtcResult> r = rtc_compile_custom(custom_body, dtype, arch.c_str(), msg, nullptr, -1, (int)d);
        if (!r) {
            if (why) *why = msg;
            return bail(rtc().ok() ? GPHIP_ERR_ARG : GPHIP_ERR_UNSUPPORTED);
        }
        if (hipModuleLoadData(&h->cmod, r->code.data()) != hipSuccess ||
            hipModuleGetFunction(&h->f_cbuild, h->cmod, r->build.c_str()) != hipSuccess ||
            hipModuleGetFunction(&h->f_cdiag, h->cmod, r->diag.c_str()) != hipSuccess ||
            hipModuleGetFunction(&h->f_cprep, h->cmod, r->prep.c_str()) != hipSuccess) {
            if (why) *why = "loading the compiled covariance function failed";
            return bail(GPHIP_ERR_HIP);
        }
    }
    apply_env_options(h);
    *out = h;
    return GPHIP_OK;
}

/* why the last gphip_create_custom* of this thread failed (compiler log of the covariance function, ..) */
const char* gphip_create_error(void) { return g_create_error.c_str(); }

/* Compiles a covariance function exactly as gphip_create_custom would, without a handle and without a device (hiprtc
 * cross-compiles): a caller can validate user input early, a deployment can check that the library found hiprtc and carries
 * its kernel text, and the code object lands in the per-process cache the next gphip_create_custom of the same function hits. */
int gphip_custom_compile_d(const char* body, int dtype, const char* arch, int grad_nparams, int d, int* cache_hit) {
    g_create_error.clear();
    if (d < 0) { g_create_error = "negative input dimension"; return GPHIP_ERR_ARG; }
    if (!body || !*body || (dtype != 64 && dtype != 32)) { g_create_error = "null / empty function body or bad dtype"; return GPHIP_ERR_ARG; }
    bool hit = false;
    std::string msg;
    if (grad_nparams == 0 || grad_nparams > 64) { g_create_error = "the gradient program takes 1 .. 64 hyper-parameters"; return GPHIP_ERR_ARG; }
    const std::shared_ptr<const RtcResult> r = rtc_compile_custom(body, dtype, (arch && *arch) ? arch : "gfx950", msg, &hit, grad_nparams < 0 ? -1 : grad_nparams, d);
    if (cache_hit) *cache_hit = hit ? 1 : 0;
    if (!r) {
        g_create_error = msg;
        return rtc().ok() ? GPHIP_ERR_ARG : GPHIP_ERR_UNSUPPORTED;
    }
    return GPHIP_OK;
}

/* the dimension-generic program (d = 0): what a handle with d > 32 inputs runs */
int gphip_custom_compile(const char* body, int dtype, const char* arch, int grad_nparams, int* cache_hit) {
    return gphip_custom_compile_d(body, dtype, arch, grad_nparams, 0, cache_hit);
}

static int group_create(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id, int dtype,
                        const int* devices, const int* ranks, int nlocal, int world, const void* id128, gphip_handle* out);

int gphip_create_custom_devices(const void* X, const void* y, int64_t N, int64_t d, const char* body, int nparams, int mean_id,
                                int dtype, const int* devices, int ndev, gphip_handle* out) {
    g_create_error.clear();
    if (!out) return GPHIP_ERR_ARG;
    *out = nullptr;
    if (!body || !*body || nparams < 0 || nparams > 4096) { g_create_error = "null / empty function body or bad parameter count"; return GPHIP_ERR_ARG; }
    if (ndev < 0 || (ndev > 0 && !devices)) return GPHIP_ERR_ARG;
    g_pending_body = body; g_pending_ncp = nparams;
    int rc;
    if (ndev <= 1) {
        rc = create_ctx(X, y, N, d, GPHIP_KERNEL_CUSTOM, mean_id, dtype, ndev == 1 ? devices[0] : -1, out);
    } else {
        std::vector<int> ranks((size_t)ndev);
        for (int i = 0; i < ndev; ++i) ranks[(size_t)i] = i;
        rc = group_create(X, y, N, d, GPHIP_KERNEL_CUSTOM, mean_id, dtype, devices, ranks.data(), ndev, ndev, nullptr, out);
    }
    g_pending_body = nullptr; g_pending_ncp = 0;
    return rc;
}

int gphip_create_custom(const void* X, const void* y, int64_t N, int64_t d, const char* body, int nparams, int mean_id,
                        int dtype, int device, gphip_handle* out) {
    return gphip_create_custom_devices(X, y, N, d, body, nparams, mean_id, dtype, &device, device < 0 ? 0 : 1, out);
}

int gphip_create_custom_rank(const void* X, const void* y, int64_t N, int64_t d, const char* body, int nparams, int mean_id,
                             int dtype, int device, int rank, int world, const void* id128, gphip_handle* out) {
    g_create_error.clear();
    if (!out) return GPHIP_ERR_ARG;
    *out = nullptr;
    if (!body || !*body || nparams < 0 || nparams > 4096) { g_create_error = "null / empty function body or bad parameter count"; return GPHIP_ERR_ARG; }
    if (world < 1 || rank < 0 || rank >= world || !id128) return GPHIP_ERR_ARG;
    g_pending_body = body; g_pending_ncp = nparams;
    const int rc = group_create(X, y, N, d, GPHIP_KERNEL_CUSTOM, mean_id, dtype, &device, &rank, 1, world, id128, out);
    g_pending_body = nullptr; g_pending_ncp = 0;
    return rc;
}

// devices/ndev: NULL/0 = the current device; one ordinal = that device; several = a multi-device handle
// (one context per listed device; a repeated ordinal gives several virtual ranks on
// one GPU -- how the sharded schedule is exercised on a single-GPU box).
int gphip_create(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id,
                 int dtype, const int* devices, int ndev, gphip_handle* out) {
    if (!out) return GPHIP_ERR_ARG;
    *out = nullptr;
    if (ndev < 0 || (ndev > 0 && !devices)) return GPHIP_ERR_ARG;
    if (ndev <= 1) return create_ctx(X, y, N, d, kernel_id, mean_id, dtype, ndev == 1 ? devices[0] : -1, out);
    if (kernel_id == GPHIP_KERNEL_NULL) return GPHIP_ERR_UNSUPPORTED;   // nothing to shard: K is diagonal
    std::vector<int> ranks(ndev);
    for (int i = 0; i < ndev; ++i) ranks[i] = i;
    return group_create(X, y, N, d, kernel_id, mean_id, dtype, devices, ranks.data(), ndev, ndev, nullptr, out);
}

int gphip_comm_unique_id(void* id128) {
    if (!id128) return GPHIP_ERR_ARG;
    const RcclApi& api = rccl();
    if (!api.ok()) return GPHIP_ERR_UNSUPPORTED;
    NcclUniqueId id;
    if (api.GetUniqueId(&id) != 0) return GPHIP_ERR_HIP;
    memcpy(id128, id.internal, GPHIP_COMM_ID_BYTES);
    return GPHIP_OK;
}

int gphip_create_rank(const void* X, const void* y, int64_t N, int64_t d, int kernel_id, int mean_id, int dtype,
                      int device, int rank, int world, const void* id128, gphip_handle* out) {
    if (!out) return GPHIP_ERR_ARG;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !id128) return GPHIP_ERR_ARG;
    if (kernel_id == GPHIP_KERNEL_NULL) return GPHIP_ERR_UNSUPPORTED;
    return group_create(X, y, N, d, kernel_id, mean_id, dtype, &device, &rank, 1, world, id128, out);
}

int gphip_destroy(gphip_handle h) {
    if (!h) return GPHIP_OK;
    (void)hipSetDevice(h->device);
    if (h->cstream) (void)hipStreamSynchronize(h->cstream);
    if (h->pstream) (void)hipStreamSynchronize(h->pstream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    group_destroy(h);                          // communicators + the other members of a multi-device handle
    (void)hipSetDevice(h->device);
    for (void* pk : h->packed) (void)hipFree(pk);
    (void)hipFree(h->dOwn); (void)hipFree(h->dDistAdj); (void)hipFree(h->dZ);
    (void)hipFree(h->dScal8); (void)hipFree(h->drain_buf);
    (void)hipFree(h->dKss);
    if (h->cmod) (void)hipModuleUnload(h->cmod);
    if (h->cstream) (void)hipStreamDestroy(h->cstream);
    free_slots(h);
    (void)hipFree(h->dXt); (void)hipFree(h->dY); (void)hipFree(h->dExp2); (void)hipFree(h->dCentre);
    if (h->cgmod) (void)hipModuleUnload(h->cgmod);
    (void)hipFree(h->dV); (void)hipFree(h->dXsT); (void)hipFree(h->dXsS); (void)hipFree(h->dMean);
    (void)hipFree(h->dVar); (void)hipFree(h->dAlpha); (void)hipFree(h->dGacc); (void)hipFree(h->dKinv);
    (void)hipFree(h->dTrsvX); (void)hipFree(h->dTrsvP); (void)hipFree(h->dRows); (void)hipFree(h->dColSig); (void)hipFree(h->dGpart); (void)hipFree(h->dW64s);
    (void)hipFree(h->dXsS2); (void)hipFree(h->dPwMeanT); (void)hipFree(h->dPwNugT);
    (void)hipFree(h->dNullMu); (void)hipFree(h->dNullOut); (void)hipFree(h->dPart);
    for (auto e : h->pool) (void)hipEventDestroy(e);
    for (auto e : h->sync_events) (void)hipEventDestroy(e);
    if (h->own_streams) {
        if (h->pstream) (void)hipStreamDestroy(h->pstream);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return GPHIP_OK;
}

int gphip_num_params(gphip_handle h, int* p) {
    if (!h || !p) return GPHIP_ERR_ARG;
    *p = h->p;
    return GPHIP_OK;
}

int gphip_loglik(gphip_handle h, const double* theta, int p, double* out, int* info) {
    return eval_batch(h, theta, 1, p, out, nullptr, info);
}

int gphip_loglik_parts(gphip_handle h, const double* theta, int p, double* out, double* parts, int* info) {
    return eval_batch(h, theta, 1, p, out, parts, info);
}

int gphip_loglik_batch(gphip_handle h, const double* Theta, int B, int p, double* out, int* info) {
    return eval_batch(h, Theta, B, p, out, nullptr, info);
}

// log-likelihood and its gradient with respect to theta (same layout as theta).  One factorisation,
// then K^-1 is streamed through the scratch block up to 8 GiB of rows at a time (forward + backward
// substitution of identity rows) and contracted against dK/dtheta on the fly.
int gphip_loglik_grad(gphip_handle h, const double* theta, int p, double* out, double* grad, int* info) {
    if (!h || !theta || !out || !grad || !info) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (h->kernel_id == GPHIP_KERNEL_NULL) {
        std::lock_guard<std::recursive_mutex> lk(h->mu);
        if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length");
        HIPCHK(hipSetDevice(h->device));
        return null_kernel_batch(h, theta, 1, out, nullptr, info, grad);
    }
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->custom) {
        if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length");
        // One factorisation: the function's text instantiated with forward-mode dual numbers inside the gradient reduction
        // (custom_grad_kernel), compiled on first use.  A body that does not compile that way (a math function gp_dual.h
        // does not differentiate, intermediates of a fixed scalar type, ..) keeps the difference route below.
        constexpr int CGRAD_MAX_NCP = 64;
        if (h->cgrad_state == 0 && h->custom_grad && h->ncp >= 1 && h->ncp <= CGRAD_MAX_NCP) {
            std::string msg;
            const std::shared_ptr<const RtcResult> r = rtc_compile_custom(h->custom_body, h->dtype, h->arch.c_str(), msg, nullptr, h->ncp, h->d);
            h->cgrad_state = -1;
            HIPCHK(hipSetDevice(h->device));
            if (r && hipModuleLoadData(&h->cgmod, r->code.data()) == hipSuccess &&
                hipModuleGetFunction(&h->f_cgrad, h->cgmod, r->grad.c_str()) == hipSuccess)
                h->cgrad_state = 1;
            else
                (void)hipGetLastError();
        }
    }
    h->grad_analytic = 0;
    if (h->custom && !(h->custom_grad && h->cgrad_state == 1)) {
        // central differences of the likelihood, all 2 p + 1 points as ONE batched evaluation.  Step eps^(1/3) max(|theta_k|,
        // 1e-2) in the handle's arithmetic (fp64 6e-6, fp32 5e-3): the truncation error (h^2 / 6 times the third derivative)
        // and the rounding error eps_ll |ll| / h balance there; expect ~1e-6 relative in fp64 at moderate N, ~1e-2 in fp32.
        const int B = 2 * p + 1;
        std::vector<double> Th((size_t)B * p), ll((size_t)B, 0.0), step((size_t)p);
        std::vector<int> inf((size_t)B, 0);
        for (int r = 0; r < B; ++r) memcpy(&Th[(size_t)r * p], theta, (size_t)p * 8);
        const double rel = h->dtype == 64 ? 6.0e-6 : 5.0e-3;
        for (int k = 0; k < p; ++k) {
            step[(size_t)k] = rel * std::max(std::fabs(theta[k]), 1e-2);
            Th[(size_t)(1 + 2 * k) * p + k] += step[(size_t)k];
            Th[(size_t)(2 + 2 * k) * p + k] -= step[(size_t)k];
        }
        const int rc = eval_batch(h, Th.data(), B, p, ll.data(), nullptr, inf.data());
        if (rc) return rc;
        *out = ll[0];
        *info = inf[0];
        for (int k = 0; k < p; ++k)
            grad[k] = (inf[0] == 0 && inf[(size_t)1 + 2 * k] == 0 && inf[(size_t)2 + 2 * k] == 0)
                          ? (ll[(size_t)1 + 2 * k] - ll[(size_t)2 + 2 * k]) / (2.0 * step[(size_t)k]) : std::nan("");
        return GPHIP_OK;
    }
    // (the gradient reductions keep (d + 1) point tiles -- general form: 4 d + 1 -- in LDS up to KB_LDS_MAXD dimensions; beyond,
    //  launch_grad reads the points from global memory in windows of 32 length-scale derivatives)
    double parts[2] = {0, 0};
    HIPCHK(hipSetDevice(h->device));
    const int64_t N = h->N, Npad = h->Npad, d = h->d;
    int rc = GPHIP_OK;
    // potri route when U (Npad x Npad scratch) and the lower tiles of K^-1 both fit in a quarter of the HBM
    // that is free right now; otherwise K^-1 is streamed in row blocks through forward + backward substitution
    bool potri = h->grad_potri != 0;
    if (potri && !(h->dKinv && h->vcap >= Npad + GRAD_LD_PAD)) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        const size_t need = (size_t)(h->dKinv ? 1 : 2) * (Npad + GRAD_LD_PAD) * Npad * h->es;
        potri = need <= fr / 4;
    }
    if (potri) {
        rc = ensure_vbuf(h, Npad + GRAD_LD_PAD);
        if (rc == GPHIP_OK && !h->dKinv && hipMalloc(&h->dKinv, (size_t)(Npad + GRAD_LD_PAD) * Npad * h->es) != hipSuccess) {
            (void)hipGetLastError();
            h->dKinv = nullptr;
            potri = false;
        }
        if (rc != GPHIP_OK) { (void)hipGetLastError(); potri = false; }
    }
    h->want_w = true;                          // (a multi-device handle factors on its first device: the K^-1 contraction needs the whole factor)
    h->want_u = potri && h->grad_potri == 1;   // a single-launch factorisation goes on to U = L^-T in dV (launch_dataflow_inverse)
    h->u_ready = false;
    rc = eval_batch_local(h, theta, 1, p, out, parts, info);
    h->want_w = h->want_u = false;
    if (rc) { h->u_ready = false; return rc; }
    for (int i = 0; i < p; ++i) grad[i] = std::nan("");
    if (*info != 0) { h->u_ready = false; return GPHIP_OK; }
    if (!h->dAlpha) HIPCHK(hipMalloc(&h->dAlpha, (size_t)Npad * h->es));
    // general form: both terms' length scales, sf, alpha, c, sn; run-time compiled function: its ncp parameters, sn
    const size_t ngacc = std::max((size_t)2 * d + 6, (size_t)h->ncp + 1);
    if (!h->dGacc) HIPCHK(hipMalloc(&h->dGacc, ngacc * 8));
    h->ngacc = ngacc;
    int64_t MC = 0;
    if (!potri) {
        // rows of K^-1 per pass: as many as keep the scratch block within ~8 GiB (each pass runs a forward and a
        // backward substitution over all of L; few, tall passes keep their launches chip-filling)
        MC = (int64_t)((8.0 * (1 << 30)) / ((double)Npad * h->es)) / TB * TB;
        if (MC < 2048) MC = 2048;
        if (MC > Npad) MC = Npad;
        rc = ensure_vbuf(h, MC);
        while (rc == GPHIP_ERR_HIP && MC > 2048) {
            (void)hipGetLastError();
            MC = (MC / 2 + TB - 1) / TB * TB;
            rc = ensure_vbuf(h, MC);
        }
        if (rc) return rc;
    }
    h->cs = h->stream;
    rc = (potri && h->u_ready) ? DISPATCH(h, queue_alpha_from_u, h) : DISPATCH(h, queue_alpha, h);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(h->dGacc, 0, ngacc * 8, h->stream));
    if (potri) {
        rc = DISPATCH(h, queue_grad_potri, h);
        if (rc) return rc;
    } else {
        for (int64_t c0 = 0; c0 < N; c0 += MC) {
            const int64_t mc = (N - c0 < MC) ? (N - c0) : MC;
            const int64_t mpad = (mc + TB - 1) / TB * TB;
            rc = DISPATCH(h, queue_grad_chunk, h, c0, mc, mpad);
            if (rc) return rc;
        }
    }
    std::vector<double> gacc(ngacc), alpha;
    HIPCHK(hipMemcpyAsync(gacc.data(), h->dGacc, gacc.size() * 8, hipMemcpyDeviceToHost, h->stream));
    const bool u_df = potri && h->u_ready;     // U = L^-T came from an inverse launch of the dataflow kernel
    if (u_df) { rc = queue_abort_probe(h); if (rc) return rc; }
    rc = DISPATCH(h, download, h, alpha, h->dAlpha, (size_t)N, h->stream);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    harvest(h);
    if (u_df) { rc = abort_probe_verdict(h, "dataflow inverse launch timed out (set option grad_potri=2 and report)"); if (rc) return rc; }
    h->grad_analytic = 1;
    if (h->custom) {
        // theta = [p_0 .. p_{ncp-1}] sn [mu]  (accumulators: custom_grad_kernel)
        for (int m = 0; m < h->ncp; ++m) grad[m] = 0.5 * gacc[(size_t)m];
        grad[h->ncp] = gacc[(size_t)h->ncp] * theta[h->ncp];
        if (h->mean_id == GPHIP_MEAN_CONST) {
            double sum = 0.0;
            for (double v : alpha) sum += v;
            grad[h->ncp + 1] = sum;
        }
        h->fitted = true;
        h->dist_fit = false;
        stamp_fit(h);
        h->theta_fit.assign(theta, theta + p);
        h->logdet_fit = parts[0];
        h->mu_fit = h->hSlotp[2];
        h->kappa_fit = h->hSlotp[SP_KXX] + h->hSlotp[1];
        return GPHIP_OK;
    }
    // chain rule onto the theta layout [term 1: l.., (alpha), sf] [term 2] [c] sn [mu]  (accumulators: grad_reduce_*)
    int o = 0;
    auto lengths = [&](int nl, size_t base) {
        if (nl == 1) {
            double sum = 0.0;
            for (int64_t j = 0; j < d; ++j) sum += gacc[base + (size_t)j];
            grad[o] = 0.5 * sum / theta[o];                   // even in l: d/dl of f(l^2)
        } else {
            for (int64_t j = 0; j < d; ++j) grad[o + j] = 0.5 * gacc[base + (size_t)j] / theta[o + j];
        }
        o += nl;
    };
    lengths(h->nl, 0);
    if (h->has_a1) grad[o++] = 0.5 * gacc[(size_t)2 * d + 3];
    grad[o] = gacc[(size_t)d] / theta[o]; ++o;                // sf1: 1/2 * sum w (dk/dk1) k1 * 2/sf
    if (h->nl2 > 0) {
        lengths(h->nl2, (size_t)d + 2);
        if (h->has_a2) grad[o++] = 0.5 * gacc[(size_t)2 * d + 4];
        grad[o] = gacc[(size_t)2 * d + 2] / theta[o]; ++o;
    }
    if (h->ks.offset) grad[o++] = 0.5 * gacc[(size_t)2 * d + 5];
    grad[o] = gacc[(size_t)d + 1] * theta[o]; ++o;            // sn: 1/2 * tr(W) * 2 sn
    if (h->mean_id == GPHIP_MEAN_CONST) {
        double sum = 0.0;
        for (double v : alpha) sum += v;
        grad[o++] = sum;
    }
    h->fitted = true;                                         // the factor of theta is still resident (whole, on this device)
    h->dist_fit = false;
    stamp_fit(h);
    h->theta_fit.assign(theta, theta + p);
    h->logdet_fit = parts[0];
    h->mu_fit = h->hSlotp[2];
    h->kappa_fit = h->hSlotp[SP_KXX] + h->hSlotp[1];
    return GPHIP_OK;
}

int gphip_fit(gphip_handle h, const double* theta, int p, int* info) {
    if (!h || !theta || !info) return fail(h, GPHIP_ERR_ARG, "null argument");
    double out, parts[2];
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->kernel_id == GPHIP_KERNEL_NULL) {
        // null kernel Function[0] (BGP:25-27, 63-89, 156-159): K = diag(sn^2) -- nothing to factor.  "Inverse" is a
        // division by the diagonal, LogDet = N log sn^2, and prediction has k = 0 (empty sparse array, BGP:72-74)
        // and kappa = nugget only (BGP:75-80): mu* = m(x*), var* = sn^2.
        if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length for this kernel/mean");
        const double sn = theta[0], mu = (h->mean_id == GPHIP_MEAN_CONST) ? theta[1] : 0.0;
        const bool finite = std::isfinite(sn) && std::isfinite(mu);
        *info = !finite ? GPHIP_INFO_NAN : (sn * sn > 0.0 && std::isfinite(std::log(sn * sn)) ? GPHIP_INFO_OK : GPHIP_INFO_NOT_SPD);
        h->logdet_fit = (double)h->N * std::log(sn * sn);
        h->null_diag.clear();
        if (h->pw_nug_host) {                  // nugget /@ points (BGP:27): the diagonal itself
            h->null_diag.assign(h->pw_nug_host, h->pw_nug_host + h->N);
            double ld = 0.0;
            bool pos = true, fin = finite;
            for (double v : h->null_diag) { ld += std::log(std::fabs(v)); pos = pos && v > 0.0; fin = fin && std::isfinite(v); }
            h->logdet_fit = ld;
            *info = !fin ? GPHIP_INFO_NAN : (pos && std::isfinite(ld) ? GPHIP_INFO_OK : GPHIP_INFO_NOT_SPD);
        }
        h->fitted = h->null_fit = (*info == 0);
        h->dist_fit = false;
        stamp_fit(h);
        h->theta_fit.assign(theta, theta + p);
        h->mu_fit = mu;
        h->kappa_fit = sn * sn;
        return GPHIP_OK;
    }
    h->want_w = true;                          // the substitutions that follow a fit use the 128-block inverses
    const unsigned long id_before = h->group ? h->group->fit_id : 0;
    int rc = eval_batch(h, theta, 1, p, &out, parts, info);
    h->want_w = false;
    if (rc) return rc;
    const bool sharded_fit = h->group && h->group->fit_id != id_before;     // group_eval_run stamped every member itself
    if (h->dist_fit) {                         // (a later gphip_solve factors locally again: it needs the same K)
        h->fit_pw_mean.assign(h->pw_mean_host ? h->pw_mean_host : nullptr, h->pw_mean_host ? h->pw_mean_host + h->N : nullptr);
        h->fit_pw_nug.assign(h->pw_nug_host ? h->pw_nug_host : nullptr, h->pw_nug_host ? h->pw_nug_host + h->N : nullptr);
    }
    h->fitted = (*info == 0);
    if (!sharded_fit) stamp_fit(h);
    h->theta_fit.assign(theta, theta + p);
    h->logdet_fit = parts[0];
    h->mu_fit = h->hSlotp[2];
    h->kappa_fit = h->hSlotp[SP_KXX] + h->hSlotp[1];
    return GPHIP_OK;
}

int gphip_logdet(gphip_handle h, double* out) {
    if (!h || !out) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (!has_fit(h)) return fail(h, GPHIP_ERR_STATE, "gphip_logdet before a successful gphip_fit");
    *out = h->logdet_fit;
    return GPHIP_OK;
}

int gphip_covariance(gphip_handle h, const double* theta, int p, double* K) {
    if (!h || !theta || !K) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    const int64_t N = h->N;
    if (h->kernel_id == GPHIP_KERNEL_NULL) {
        for (int64_t i = 0; i < N * N; ++i) K[i] = 0.0;
        for (int64_t i = 0; i < N; ++i) K[i * N + i] = theta[0] * theta[0];
        return GPHIP_OK;
    }
    int rc = ensure_slots(h, 1);
    if (rc) return rc;
    invalidate_fit(h);
    if (!stage_theta(h, 0, theta)) return fail(h, GPHIP_ERR_ARG, "non-finite or zero hyper-parameter");
    rc = copy_theta(h, 1);
    if (rc) return rc;
    h->cs = h->stream;
    DISPATCH(h, queue_build, h, 1);
    // the lower-triangle tiles of the packed workspace (tile (ti, tj) at tile_index(ti, tj, R), column-major inside)
    std::vector<double> tmp;
    rc = DISPATCH(h, download, h, tmp, h->dA, (size_t)h->slot_elems, h->stream);
    if (rc) return rc;
    harvest(h);
    for (int64_t j = 0; j < N; ++j)
        for (int64_t i = j; i < N; ++i) {
            const size_t off = (size_t)tile_index((int)(i / TB), (int)(j / TB), (int)h->R) * TS + (size_t)(j % TB) * TB + (size_t)(i % TB);
            K[i * N + j] = K[j * N + i] = tmp[off];
        }
    return GPHIP_OK;
}

// compiledKandKappa (BGP:91-124; null kernel BGP:63-89): k = Table[kernel[i, j], {i, points1}, {j, points2}]
// (row-major N x M: rows = training points, columns = test points) and kappa_j = kernel[x*_j, x*_j] +
// nugget[x*_j].  Needs no fit; leaves the handle un-fitted (the scaled inputs of slot 0 are overwritten).
int gphip_cross_covariance(gphip_handle h, const double* theta, int p, const void* Xs, int64_t M, double* k, double* kappa) {
    if (!h || !theta || !Xs || !k || !kappa) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length");
    if (M < 1) return fail(h, GPHIP_ERR_DIM, "M < 1");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    const int64_t N = h->N, d = h->d;
    if (h->kernel_id == GPHIP_KERNEL_NULL) {   // k = SparseArray[{}, {N, M}], kappa = nugget (BGP:72-80)
        for (int64_t i = 0; i < N * M; ++i) k[i] = 0.0;
        for (int64_t t = 0; t < M; ++t) kappa[t] = theta[0] * theta[0];
        return GPHIP_OK;
    }
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_slots(h, 1);
    if (rc) return rc;
    invalidate_fit(h);
    if (!stage_theta(h, 0, theta)) return fail(h, GPHIP_ERR_ARG, "non-finite or zero hyper-parameter");
    rc = copy_theta(h, 1);
    if (rc) return rc;
    h->cs = h->stream;
    DISPATCH(h, queue_scale_train, h);
    const double* X = static_cast<const double*>(Xs);
    const int64_t MC = 2048;
    rc = ensure_vbuf(h, M < MC ? (M + TB - 1) / TB * TB : MC);
    if (rc) return rc;
    std::vector<double> xt, v;
    for (int64_t m0 = 0; m0 < M; m0 += MC) {
        const int64_t mc = (M - m0 < MC) ? (M - m0) : MC;
        const int64_t mpad = (mc + TB - 1) / TB * TB;
        xt.assign((size_t)d * mpad, 0.0);
        for (int64_t i = 0; i < mc; ++i)
            for (int64_t j = 0; j < d; ++j) xt[(size_t)j * mpad + i] = X[(m0 + i) * d + j];
        note_test_range(h, xt, mc, mpad);
        rc = DISPATCH(h, upload, h, h->dXsT, xt, h->stream);
        if (rc) return rc;
        DISPATCH(h, queue_cross, h, mc, mpad, 1);
        rc = DISPATCH(h, download, h, v, h->dV, (size_t)mpad * (size_t)N, h->stream);   // V(t, j) at j*mpad + t
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        harvest(h);
        for (int64_t j = 0; j < N; ++j)
            for (int64_t t = 0; t < mc; ++t) k[j * M + m0 + t] = v[(size_t)j * mpad + t];
        if (h->custom) {                           // kappa_t = k(x*_t, x*_t) + nugget: a function of the point (device)
            rc = queue_custom_kss(h, mc, mpad, 1);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(kappa + m0, h->dKss, (size_t)mc * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int64_t t = 0; t < mc; ++t) kappa[m0 + t] += h->hSlotp[1];
        }
    }
    if (!h->custom)
        for (int64_t t = 0; t < M; ++t) kappa[t] = h->hSlotp[SP_KXX] + h->hSlotp[1];
    return GPHIP_OK;
}

// world = ranks of the job this handle belongs to (1 for a plain handle), nlocal = ranks in this process,
// *comm = "none" / "device copies" / "rccl (..)" (owned by the library)
int gphip_comm_info(gphip_handle h, int* world, int* nlocal, const char** comm) {
    if (!h) return GPHIP_ERR_ARG;
    if (world) *world = h->group ? h->group->world : 1;
    if (nlocal) *nlocal = h->group ? (int)h->group->members.size() : 1;
    if (comm) *comm = h->group ? h->group->comm_name.c_str() : "none";
    return GPHIP_OK;
}

// the Listable form of compiledCovarianceMatrix (BGP:59: a B x p matrix of thetas gives B matrices): K row-major
// B x N x N; a theta that cannot be used (non-finite / zero length scale) fails the whole call like gphip_covariance
int gphip_covariance_batch(gphip_handle h, const double* Theta, int B, int p, double* K) {
    if (!h || !Theta || !K) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (B < 1) return fail(h, GPHIP_ERR_DIM, "B < 1");
    for (int b = 0; b < B; ++b) {
        const int rc = gphip_covariance(h, Theta + (size_t)b * p, p, K + (size_t)b * h->N * h->N);
        if (rc) return rc;
    }
    return GPHIP_OK;
}

static int predict_local(gphip_handle h, const void* Xs, int64_t M, double* mean, double* var);
static int predict_streamed(gphip_handle h, const void* Xs, int64_t M, double* mean, double* var);

int gphip_predict(gphip_handle h, const void* Xs, int64_t M, double* mean, double* var) {
    if (!h || !Xs || !mean || !var) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (M < 1) return fail(h, GPHIP_ERR_DIM, "M < 1");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (!has_fit(h)) return fail(h, GPHIP_ERR_STATE, "gphip_predict before a successful gphip_fit");
    if (h->null_fit) {                         // null kernel (BGP:63-89): k = 0, kappa = nugget
        for (int64_t t = 0; t < M; ++t) {
            mean[t] = h->pw_mean_test ? h->pw_mean_test[t] : h->mu_fit;
            var[t] = h->pw_nug_test ? h->pw_nug_test[t] : h->kappa_fit;
        }
        return GPHIP_OK;
    }
    // the factor of the fitted theta is spread over the ranks (sharded fit, replicate_factor = 0): its panels are
    // streamed through every rank once more, each rank substituting ITS test points -- a collective call
    if (h->dist_fit) return predict_streamed(h, Xs, M, mean, var);
    // multi-device handle whose members all hold the factor (sharded fit, replicate_factor = 1): test points shard, no collective
    if (h->group && h->group->members.size() > 1 && M >= 2 * TB * (int64_t)h->group->members.size()) {
        gphip_group* g = h->group;
        bool all = true;
        for (gphip_ctx* m : g->members) all = all && has_fit(m) && m->fit_id == g->fit_id && m->theta_fit == h->theta_fit;
        if (all) {
            const int64_t nl = (int64_t)g->members.size(), d = h->d;
            const double* X = static_cast<const double*>(Xs);
            const double *pm = h->pw_mean_test, *pn = h->pw_nug_test;
            const int rc = group_parallel(h, [&](int i) {
                const int64_t m0 = M * i / nl, m1 = M * (i + 1) / nl;
                gphip_ctx* m = g->members[(size_t)i];
                m->pw_mean_test = pm ? pm + m0 : nullptr;      // each member's shard of m(x*), nugget(x*)
                m->pw_nug_test = pn ? pn + m0 : nullptr;
                const int c = predict_local(m, X + m0 * d, m1 - m0, mean + m0, var + m0);
                if (m != h) m->pw_mean_test = m->pw_nug_test = nullptr;
                return c;
            });
            h->pw_mean_test = pm; h->pw_nug_test = pn;
            return rc;
        }
    }
    return predict_local(h, Xs, M, mean, var);
}

static int predict_local(gphip_handle h, const void* Xs, int64_t M, double* mean, double* var) {
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    const double* X = static_cast<const double*>(Xs);
    const int64_t d = h->d;
    // test points per chunk: as many as keep V (chunk x Npad) within ~8 GiB -- every chunk re-runs the whole
    // substitution over L, and its per-block launches only fill the chip when the chunk has many row tiles
    int64_t MC = (int64_t)((8.0 * (1 << 30)) / ((double)h->Npad * h->es)) / TB * TB;
    if (MC < 2048) MC = 2048;
    if (MC > 32768) MC = 32768;
    if (M < MC) MC = (M + TB - 1) / TB * TB;
    int rc = ensure_vbuf(h, MC);
    while (rc == GPHIP_ERR_HIP && MC > 2048) {               // not enough free HBM next to the factor: smaller chunks
        (void)hipGetLastError();
        MC = (MC / 2 + TB - 1) / TB * TB;
        rc = ensure_vbuf(h, MC);
    }
    if (rc) return rc;
    h->cs = h->stream;
    std::vector<double> xt;
    for (int64_t m0 = 0; m0 < M; m0 += MC) {
        const int64_t mc = (M - m0 < MC) ? (M - m0) : MC;
        const int64_t mpad = (mc + TB - 1) / TB * TB;
        xt.assign((size_t)d * mpad, 0.0);
        for (int64_t i = 0; i < mc; ++i)
            for (int64_t j = 0; j < d; ++j) xt[(size_t)j * mpad + i] = X[(m0 + i) * d + j];
        note_test_range(h, xt, mc, mpad);
        rc = DISPATCH(h, upload, h, h->dXsT, xt, h->stream);
        if (rc) return rc;
        rc = upload_pw_test(h, 0, 1, m0, mc, mpad);
        if (rc) return rc;
        DISPATCH(h, queue_cross, h, mc, mpad, 1);
        ensure_w64(h);
        const bool dfp = df_forward_ok(h, mpad);
        if (dfp) launch_dataflow_inverse<double, 64>(h, mpad);
        else DISPATCH(h, queue_forward_rows, h, mpad, 1);
        DISPATCH(h, queue_predict_reduce, h, mc, mpad, 1);
        HIPCHK(hipMemcpyAsync(mean + m0, h->dMean, (size_t)mc * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(var + m0, h->dVar, (size_t)mc * 8, hipMemcpyDeviceToHost, h->stream));
        if (dfp) { rc = queue_abort_probe(h); if (rc) return rc; }
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
        harvest(h);
        if (dfp) { rc = abort_probe_verdict(h, "dataflow forward substitution timed out (set option predict_df=0 and report)"); if (rc) return rc; }
    }
    return GPHIP_OK;
}

// Prediction from a DISTRIBUTED factor (sharded fit with replicate_factor = 0: every rank holds only its own outer panels).
// The test points are dealt to the local ranks; then the factor's panels are broadcast once more, in order, through the
// rotating receive buffers, and every rank runs the forward substitution of ITS test points panel by panel as the panels
// arrive (the 128-block inverses of received diagonal blocks are rebuilt on arrival, z = L^-1 r is gathered from the rhs
// tile row of each panel).  Traffic: the factor once per pass (N^2/2 elements received per rank); a pass handles up to
// ~8 GiB of V per rank.  COLLECTIVE for rank handles in separate processes: every rank calls gphip_predict (each with its
// own test points, any count >= 1); the number of passes is agreed on with one all-reduce.
static int predict_streamed(gphip_handle h, const void* Xs, int64_t M, double* mean, double* var) {
    gphip_group* g = h->group;
    if (!g) return fail(h, GPHIP_ERR_STATE, "distributed fit without a group");
    if (g->broken) return fail(h, GPHIP_ERR_STATE, "a collective call failed earlier: this multi-device handle can no longer shard");
    const int nl = (int)g->members.size(), W = g->world;
    const double* X = static_cast<const double*>(Xs);
    const int64_t d = h->d;
    int nouter = 0;
    int rc = gphip_dist_num_panels(h, &nouter);
    if (rc) return rc;
    // COLLECTIVE from here on (one all-reduce, then passes x nouter broadcasts that every rank of the job issues): nothing
    // rank-local returns before the all-reduce -- it carries a "this rank cannot" flag, so all ranks bail out TOGETHER --
    // and after it a local failure only switches this process to drain mode (the remaining broadcasts still go out, through
    // a scratch buffer), exactly as in group_eval_run.
    int failed = GPHIP_OK;
    std::string failed_why;
    auto local_fail = [&](int code, const std::string& why) {
        if (code != GPHIP_OK && failed == GPHIP_OK) { failed = code; failed_why = why; }
    };
    auto soft = [&](hipError_t e, const char* what) {
        if (h->debug_fail_hip > 0 && --h->debug_fail_hip == 0) e = hipErrorUnknown;                  // fault injection (tests)
        if (e != hipSuccess) {
            (void)hipGetLastError();
            local_fail(GPHIP_ERR_HIP, std::string(what) + " failed in a streamed prediction: " + hipGetErrorString(e));
        }
        return e == hipSuccess;
    };
#define SOFT(call) soft((call), #call)
    for (gphip_ctx* m : g->members)
        if (!m->dist_fit || !has_fit(m) || m->fit_id != g->fit_id || m->theta_fit != h->theta_fit || m->lay_panel != m->panel)
            local_fail(GPHIP_ERR_STATE, "the distributed factor is gone (another call reused the buffers, or \"panel\" changed): fit again");
    // test points -> local ranks (contiguous blocks; few points: the first rank alone), chunks of <= ~8 GiB of V
    std::vector<int64_t> lo((size_t)nl + 1, M);
    lo[0] = 0;
    if (nl > 1 && M >= 2 * TB * (int64_t)nl)
        for (int i = 1; i < nl; ++i) lo[(size_t)i] = M * i / nl;
    std::vector<int64_t> MC((size_t)nl, TB);
    double passes = 1.0;
    for (int i = 0; i < nl && failed == GPHIP_OK; ++i) {
        gphip_ctx* m = g->members[(size_t)i];
        const int64_t Mi = lo[(size_t)i + 1] - lo[(size_t)i];
        if (Mi <= 0) continue;
        if (!SOFT(hipSetDevice(m->device))) break;
        int64_t mc = (int64_t)((8.0 * (1 << 30)) / ((double)m->Npad * m->es)) / TB * TB;
        mc = std::max<int64_t>(2048, std::min<int64_t>(mc, 32768));
        if (Mi < mc) mc = (Mi + TB - 1) / TB * TB;
        rc = ensure_vbuf(m, mc);
        while (rc == GPHIP_ERR_HIP && mc > 2048) {
            (void)hipGetLastError();
            mc = (mc / 2 + TB - 1) / TB * TB;
            rc = ensure_vbuf(m, mc);
        }
        if (rc) { local_fail(rc, m->err); break; }
        if (!m->dZ && !SOFT(hipMalloc(&m->dZ, (size_t)m->Npad * m->es))) break;
        MC[(size_t)i] = mc;
        passes = std::max(passes, (double)((Mi + mc - 1) / mc));
    }
    g->replicate = 0;                                                  // a distributed factor: owned panels + receive buffers
    g->two_hop = false;                                                // (whole panels, plain broadcasts: no agreement check precedes this call)
    if (failed == GPHIP_OK) {
        rc = group_resize_packed(h, g);
        if (rc) local_fail(rc, h->err);
    }
    if (nl < W) {                                  // ranks elsewhere: agree on the number of passes (max) AND on "everybody can"
        double v[2] = {passes, failed != GPHIP_OK ? 1.0 : 0.0};
        (void)hipSetDevice(h->device);
        if (!h->dScal8) { g->broken = true; return fail(h, GPHIP_ERR_HIP, "scalar buffer of the multi-device handle is missing"); }
        if (hipMemcpyAsync(h->dScal8, v, sizeof v, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            (void)hipGetLastError();
            local_fail(GPHIP_ERR_HIP, "uploading the pass count of a streamed prediction failed");
            (void)hipMemsetAsync(h->dScal8, 0x7f, sizeof v, h->stream);            // 1.4e306 in both entries: reads as "a rank cannot"
        }
        if (rccl().AllReduce(h->dScal8, h->dScal8, 2, NCCL_FLOAT64, NCCL_MAX, g->comms[0], h->stream) != 0) {
            g->broken = true;
            return fail(h, GPHIP_ERR_HIP, "ncclAllReduce (passes) failed");
        }
        if (hipMemcpyAsync(v, h->dScal8, sizeof v, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            (void)hipGetLastError();
            g->broken = true;                      // this rank cannot learn what its peers are about to do
            return fail(h, GPHIP_ERR_HIP, "reading back the pass count of a streamed prediction failed");
        }
        if (!(v[1] == 0.0)) {                      // some rank cannot: NO rank issues a broadcast
            h->err = failed != GPHIP_OK ? failed_why : std::string("another rank of the job could not prepare this streamed prediction");
            return failed != GPHIP_OK ? failed : GPHIP_ERR_HIP;
        }
        passes = v[0];
    } else if (failed != GPHIP_OK) {
        h->err = failed_why;
        return failed;
    }
    std::vector<double> xt;
    const double *pm0 = h->pw_mean_test, *pn0 = h->pw_nug_test;      // (member 0 IS the public handle: keep the call's pointers)
    struct Restore { gphip_ctx* h; const double *a, *b; ~Restore() { h->pw_mean_test = a; h->pw_nug_test = b; } } restore{h, pm0, pn0};
    auto member_fail = [&](gphip_ctx* m, int code) { if (code) local_fail(code, m->err); };
    for (int pass = 0; pass < (int)passes; ++pass) {
        std::vector<int64_t> c0((size_t)nl, 0), mcv((size_t)nl, 0), mpadv((size_t)nl, 0);
        for (int i = 0; i < nl && failed == GPHIP_OK; ++i) {           // k* of this pass's chunk on every rank that has one
            gphip_ctx* m = g->members[(size_t)i];
            m->sync_used = 0;
            const int64_t a = lo[(size_t)i] + (int64_t)pass * MC[(size_t)i], b = std::min(lo[(size_t)i + 1], a + MC[(size_t)i]);
            if (b <= a) continue;
            c0[(size_t)i] = a; mcv[(size_t)i] = b - a; mpadv[(size_t)i] = (b - a + TB - 1) / TB * TB;
            const int64_t mc = b - a, mpad = mpadv[(size_t)i];
            if (!SOFT(hipSetDevice(m->device))) break;
            m->cs = m->stream;
            xt.assign((size_t)d * mpad, 0.0);
            for (int64_t r = 0; r < mc; ++r)
                for (int64_t j = 0; j < d; ++j) xt[(size_t)j * mpad + r] = X[(a + r) * d + j];
            note_test_range(m, xt, mc, mpad);
            member_fail(m, DISPATCH(m, upload, m, m->dXsT, xt, m->stream));
            m->pw_mean_test = pm0 ? pm0 + a : nullptr;
            m->pw_nug_test = pn0 ? pn0 + a : nullptr;
            if (failed == GPHIP_OK) member_fail(m, upload_pw_test(m, 0, 1, 0, mc, mpad));
            if (failed == GPHIP_OK) DISPATCH(m, queue_cross, m, mc, mpad, 1);
        }
        std::vector<std::vector<hipEvent_t>> ev_used((size_t)nl, std::vector<hipEvent_t>((size_t)nouter, nullptr));
        for (int k = 0; k < nouter; ++k) {
            const int o = k % W;
            for (int i = 0; i < nl && failed == GPHIP_OK; ++i) {       // the receive buffer's last reader: panel k-3's substitution
                gphip_ctx* m = g->members[(size_t)i];
                SOFT(hipSetDevice(m->device));
                if (k >= 3 && ev_used[(size_t)i][(size_t)k - 3]) SOFT(hipStreamWaitEvent(m->cstream, ev_used[(size_t)i][(size_t)k - 3], 0));
                if (pass == 0 && k < 3) {                              // nothing of an earlier call may still read the buffers
                    hipEvent_t e = sync_event(m);
                    if (SOFT(hipEventRecord(e, m->stream))) SOFT(hipStreamWaitEvent(m->cstream, e, 0));
                }
            }
            const int c = group_broadcast(h, g, k, (size_t)dist_panel_tiles(h, k) * TS * h->es, o, 0, failed != GPHIP_OK);
            if (c) { g->broken = true; return c; }
            const int K0 = k * h->panel, K1 = (int)std::min<int64_t>(K0 + h->panel, h->Nt);
            for (int i = 0; i < nl && failed == GPHIP_OK; ++i) {
                gphip_ctx* m = g->members[(size_t)i];
                if (mcv[(size_t)i] <= 0) continue;
                SOFT(hipSetDevice(m->device));
                hipEvent_t eb = sync_event(m);
                if (!SOFT(hipEventRecord(eb, m->cstream)) || !SOFT(hipStreamWaitEvent(m->stream, eb, 0))) break;
                const bool mine = o == g->ranks[(size_t)i];
                // base through which this panel's tiles are addressed with their global indices
                char* base = group_panel_ptr(g, i, k) - dist_panel_first(m, k) * TS * (long)m->es;
                m->cs = m->stream;
                if (m->dtype == 64) {
                    if (!mine) hipLaunchKernelGGL(trtri128_kernel<double>, dim3((unsigned)(K1 - K0), 1), dim3(256), potrf_lds<double>(),
                                                  m->stream, (const double*)base, 0l, (double*)m->dW, (int)m->Nt, K0);
                    if (pass == 0)
                        hipLaunchKernelGGL(gather_rhs_row_kernel<double>, dim3((unsigned)(((K1 - K0) * TB + 255) / 256)), dim3(256), 0,
                                           m->stream, (const double*)base, (int)m->R, 0, K1 * TB, (double*)m->dZ, 1l, K0 * TB);
                } else {
                    if (!mine) hipLaunchKernelGGL(trtri128_kernel<float>, dim3((unsigned)(K1 - K0), 1), dim3(256), potrf_lds<float>(),
                                                  m->stream, (const float*)base, 0l, (float*)m->dW, (int)m->Nt, K0);
                    if (pass == 0)
                        hipLaunchKernelGGL(gather_rhs_row_kernel<float>, dim3((unsigned)(((K1 - K0) * TB + 255) / 256)), dim3(256), 0,
                                           m->stream, (const float*)base, (int)m->R, 0, K1 * TB, (float*)m->dZ, 1l, K0 * TB);
                }
                m->ws_override = base;
                DISPATCH(m, queue_forward_panel, m, mpadv[(size_t)i], 1, K0, K1, 0, false);
                m->ws_override = nullptr;
                hipEvent_t eu = sync_event(m);
                if (SOFT(hipEventRecord(eu, m->stream))) ev_used[(size_t)i][(size_t)k] = eu;
            }
        }
        for (int i = 0; i < nl; ++i) {
            gphip_ctx* m = g->members[(size_t)i];
            (void)hipSetDevice(m->device);
            if (mcv[(size_t)i] > 0 && failed == GPHIP_OK) {
                const int64_t mc = mcv[(size_t)i], a = c0[(size_t)i];
                m->z_vector = true;
                DISPATCH(m, queue_predict_reduce, m, mc, mpadv[(size_t)i], 1);
                m->z_vector = false;
                SOFT(hipMemcpyAsync(mean + a, m->dMean, (size_t)mc * 8, hipMemcpyDeviceToHost, m->stream));
                SOFT(hipMemcpyAsync(var + a, m->dVar, (size_t)mc * 8, hipMemcpyDeviceToHost, m->stream));
            }
            SOFT(hipStreamSynchronize(m->cstream));
            SOFT(hipStreamSynchronize(m->stream));
            SOFT(hipGetLastError());
            harvest(m);
            m->pw_mean_test = m->pw_nug_test = nullptr;
        }
    }
#undef SOFT
    if (failed != GPHIP_OK) {
        h->err = failed_why;
        return failed;
    }
    return GPHIP_OK;
}

// Batched mixture prediction (BGP:343-376): every posterior sample theta_s gets
// its own slot -- K(theta_s) built and factored for all samples of a chunk in ONE batched pass, then
// k*, the forward solve and the reductions run for all slots at once.  mean/var: row-major S x M.
int gphip_predict_samples(gphip_handle h, const double* Thetas, int S, int p, const void* Xs, int64_t M,
                          double* mean, double* var, int* info) {
    if (!h || !Thetas || !Xs || !mean || !var || !info) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length");
    if (S < 1 || M < 1) return fail(h, GPHIP_ERR_DIM, "S < 1 or M < 1");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->kernel_id == GPHIP_KERNEL_NULL) {   // BGP:63-89: one Normal[m(x*), sn] per sample, no factorisation
        for (int s = 0; s < S; ++s) {
            const double sn = Thetas[(size_t)s * p], mu = (h->mean_id == GPHIP_MEAN_CONST) ? Thetas[(size_t)s * p + 1] : 0.0;
            const bool finite = std::isfinite(sn) && std::isfinite(mu);
            info[s] = !finite ? GPHIP_INFO_NAN : (sn * sn > 0.0 ? GPHIP_INFO_OK : GPHIP_INFO_NOT_SPD);
            for (int64_t t = 0; t < M; ++t) {
                mean[(size_t)s * M + t] = h->pw_mean_test ? h->pw_mean_test[(size_t)s * h->pw_test_stride + t] : mu;
                var[(size_t)s * M + t] = h->pw_nug_test ? h->pw_nug_test[(size_t)s * h->pw_test_stride + t] : sn * sn;
            }
        }
        return GPHIP_OK;
    }
    if (h->group && h->group->members.size() > 1 && S >= 2 && !h->in_group_call) {
        // posterior samples are independent units: contiguous blocks to the local devices, no collective
        gphip_group* g = h->group;
        const int nl = (int)g->members.size();
        return group_parallel(h, [&](int i) {
            const int s0 = (int)((long)S * i / nl), s1 = (int)((long)S * (i + 1) / nl);
            if (s1 <= s0) return (int)GPHIP_OK;
            gphip_ctx* m = g->members[(size_t)i];
            const double *a0 = h->pw_mean_host, *a1 = h->pw_nug_host, *a2 = h->pw_mean_test, *a3 = h->pw_nug_test;
            if (m != h) {                                                  // the member's block of every per-sample row
                m->pw_mean_host = a0 ? a0 + (size_t)s0 * h->N : nullptr;
                m->pw_nug_host = a1 ? a1 + (size_t)s0 * h->N : nullptr;
                m->pw_mean_test = a2 ? a2 + (size_t)s0 * h->pw_test_stride : nullptr;
                m->pw_nug_test = a3 ? a3 + (size_t)s0 * h->pw_test_stride : nullptr;
                m->pw_test_stride = h->pw_test_stride;
            }
            m->in_group_call = true;
            const int c = gphip_predict_samples(m, Thetas + (size_t)s0 * p, s1 - s0, p, Xs, M, mean + (size_t)s0 * M,
                                                var + (size_t)s0 * M, info + s0);
            m->in_group_call = false;
            if (m != h) m->pw_mean_host = m->pw_nug_host = m->pw_mean_test = m->pw_nug_test = nullptr;
            return c;
        });
    }
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_slots(h, S);
    if (rc) return rc;
    invalidate_fit(h);
    const double* X = static_cast<const double*>(Xs);
    const int64_t d = h->d;
    std::vector<double> xt, hm, hv, scratch_out(1), scratch_parts;
    for (int s0 = 0; s0 < S; s0 += h->slots) {
        const int nb = (S - s0 < h->slots) ? (S - s0) : h->slots;
        std::vector<double> ll(nb);
        h->want_w = true;
        rc = eval_chunk(h, Thetas + (size_t)s0 * p, nb, ll.data(), nullptr, info + s0, s0);   // build + factor, kept
        h->want_w = false;
        if (rc) return rc;
        // test-point chunk so that the nb V blocks stay within ~8 GiB
        int64_t mcap = (int64_t)((8.0 * (1 << 30)) / ((double)nb * h->Npad * h->es)) / TB * TB;
        if (mcap < TB) mcap = TB;
        if (mcap > 2048) mcap = 2048;
        const int64_t MC = (M < mcap) ? (M + TB - 1) / TB * TB : mcap;
        rc = ensure_vbuf(h, (int64_t)nb * MC);
        if (rc) return rc;
        h->cs = h->stream;
        for (int64_t m0 = 0; m0 < M; m0 += MC) {
            const int64_t mc = (M - m0 < MC) ? (M - m0) : MC;
            const int64_t mpad = (mc + TB - 1) / TB * TB;
            xt.assign((size_t)d * mpad, 0.0);
            for (int64_t i = 0; i < mc; ++i)
                for (int64_t j = 0; j < d; ++j) xt[(size_t)j * mpad + i] = X[(m0 + i) * d + j];
            note_test_range(h, xt, mc, mpad);
        rc = DISPATCH(h, upload, h, h->dXsT, xt, h->stream);
            if (rc) return rc;
            rc = upload_pw_test(h, s0, nb, m0, mc, mpad);
            if (rc) return rc;
            DISPATCH(h, queue_cross, h, mc, mpad, nb);
            // few test points per sample: the forward substitutions of ALL samples as ONE dataflow launch (slot = sample) instead
            // of two launches per tile column (samples_forward_df)
            const bool dff = samples_forward_df(h, nb, mpad);
            if (dff) launch_dataflow_inverse<double, 64>(h, mpad, false, nb, h->dW64s);
            else DISPATCH(h, queue_forward_rows, h, mpad, nb);
            if (dff) { rc = queue_abort_probe(h); if (rc) return rc; }
            DISPATCH(h, queue_predict_reduce, h, mc, mpad, nb);
            hm.resize((size_t)nb * mpad);
            hv.resize((size_t)nb * mpad);
            HIPCHK(hipMemcpyAsync(hm.data(), h->dMean, hm.size() * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(hv.data(), h->dVar, hv.size() * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            HIPCHK(hipGetLastError());
            harvest(h);
            if (dff) { rc = abort_probe_verdict(h, "dataflow forward substitution timed out (set option predict_df=0 and report)"); if (rc) return rc; }
            for (int s = 0; s < nb; ++s)
                for (int64_t t = 0; t < mc; ++t) {
                    mean[(size_t)(s0 + s) * M + m0 + t] = hm[(size_t)s * mpad + t];
                    var[(size_t)(s0 + s) * M + m0 + t] = hv[(size_t)s * mpad + t];
                }
        }
    }
    return GPHIP_OK;
}

// ---- point-dependent nugget and mean (BGP:37 nugget[points[[i]]], BGP:171/300 meanFunction /@ inputData, BGP:113 nugget at
// the test points, BGP:408 meanFunction /@ inputs): the host evaluates the two functions for the theta(s) of the call and
// hands the VALUES over; a null pointer keeps the constant form (sn^2, mu) read from theta.  The arrays only have to
// live for the duration of the call.
namespace {
struct PwScope {                               // installs the call's arrays on the handle, removes them on every exit path
    gphip_ctx* h;
    PwScope(gphip_ctx* h_, const double* mt, const double* nt, const double* ms, const double* ns, long stride) : h(h_) {
        h->pw_mean_host = mt; h->pw_nug_host = nt; h->pw_mean_test = ms; h->pw_nug_test = ns; h->pw_test_stride = stride;
    }
    ~PwScope() {
        h->pw_mean_host = h->pw_nug_host = h->pw_mean_test = h->pw_nug_test = nullptr;
        h->pw_mean_on = h->pw_nug_on = false;
    }
};
}  // namespace

int gphip_loglik_batch_pw(gphip_handle h, const double* Theta, int B, int p, const double* mean_train, const double* nugget_train,
                          double* out, int* info) {
    if (!h) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    PwScope pw(h, mean_train, nugget_train, nullptr, nullptr, 0);
    return eval_batch(h, Theta, B, p, out, nullptr, info);
}

int gphip_fit_pw(gphip_handle h, const double* theta, int p, const double* mean_train, const double* nugget_train, int* info) {
    if (!h) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    PwScope pw(h, mean_train, nugget_train, nullptr, nullptr, 0);
    return gphip_fit(h, theta, p, info);
}

int gphip_predict_pw(gphip_handle h, const void* Xs, int64_t M, const double* mean_test, const double* nugget_test, double* mean,
                     double* var) {
    if (!h) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    PwScope pw(h, nullptr, nullptr, mean_test, nugget_test, (long)M);
    return gphip_predict(h, Xs, M, mean, var);
}

int gphip_predict_samples_pw(gphip_handle h, const double* Thetas, int S, int p, const double* mean_train, const double* nugget_train,
                             const void* Xs, int64_t M, const double* mean_test, const double* nugget_test, double* mean, double* var,
                             int* info) {
    if (!h) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    PwScope pw(h, mean_train, nugget_train, mean_test, nugget_test, (long)M);
    return gphip_predict_samples(h, Thetas, S, p, Xs, M, mean, var, info);
}

// out = K^-1 rhs = L^-T (L^-1 rhs): right-hand sides ride as ROWS of the scratch block V
// (V(t, j) = rhs_t[j]), forward pass V <- V L^-T, backward pass V <- V L^-1.
int gphip_solve(gphip_handle h, const double* rhs, int64_t nrhs, double* out) {
    if (!h || !rhs || !out) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (nrhs < 1) return fail(h, GPHIP_ERR_DIM, "nrhs < 1");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (!has_fit(h)) return fail(h, GPHIP_ERR_STATE, "gphip_solve before a successful gphip_fit");
    if (h->dist_fit) {
        // the factor is spread over the ranks: "Inverse" is a parity helper for small systems, so the first device simply
        // factors the fitted theta again on its own (in separate processes: every rank does) and substitutes locally
        const std::vector<double> th = h->theta_fit;
        double ll, parts[2];
        int info = 0;
        h->want_w = true;
        const double *pm = h->pw_mean_host, *pn = h->pw_nug_host;
        h->pw_mean_host = h->fit_pw_mean.empty() ? nullptr : h->fit_pw_mean.data();
        h->pw_nug_host = h->fit_pw_nug.empty() ? nullptr : h->fit_pw_nug.data();
        int rc = eval_batch_local(h, th.data(), 1, (int)th.size(), &ll, parts, &info);
        h->pw_mean_host = pm; h->pw_nug_host = pn;
        h->want_w = false;
        if (rc) return rc;
        if (info != 0) return fail(h, GPHIP_ERR_STATE, "the fitted theta no longer factors");
        h->fitted = true;                      // now a LOCAL fit of the same theta (logdet_fit / mu_fit / kappa_fit unchanged)
        stamp_fit(h);
        h->dist_fit = false;
    }
    if (h->null_fit) {                         // "Inverse" -> Function[Divide[#, matrixDiagonal]]  (BGP:156-159)
        for (int64_t i = 0; i < nrhs * h->N; ++i)
            out[i] = rhs[i] / (h->null_diag.empty() ? h->kappa_fit : h->null_diag[(size_t)(i % h->N)]);
        return GPHIP_OK;
    }
    HIPCHK(hipSetDevice(h->device));
    const int64_t N = h->N, Npad = h->Npad, MC = 2048;
    // (a non-finite right-hand side keeps the GEMM-shaped substitution: the single-vector launches recognise "not written yet" by
    //  an all-ones NaN pattern, which arithmetic on a NaN with that payload could reproduce -- a seconds-long wait, then an error)
    // 1 .. 4 vectors always, up to 16 from N = 12288 on (there the GEMM-shaped substitution is hundreds of launches at ~17 us of
    // host time each: five vectors at N = 16384 14.6 ms against two batches of the single-vector launches, ~2.5 ms)
    bool rhs_finite = nrhs <= TRSV_MAXR || (nrhs <= 4 * TRSV_MAXR && h->Nt >= 96);
    for (int64_t i = 0; rhs_finite && i < nrhs * N; ++i) rhs_finite = std::isfinite(rhs[i]);
    if (rhs_finite && trsv_ok(h, (int)std::min<int64_t>(nrhs, TRSV_MAXR))) {
        // two launches per batch of <= 4 right-hand sides that stream the factor once each (gp_trsv.h) instead of a 128-row GEMM substitution
        h->cs = h->stream;
        std::vector<double> b;
        std::vector<float> b32;
        for (int64_t m0 = 0; m0 < nrhs; m0 += TRSV_MAXR) {
            const int nr = (int)std::min<int64_t>(TRSV_MAXR, nrhs - m0);
            b.assign((size_t)nr * Npad, 0.0);
            for (int t = 0; t < nr; ++t) memcpy(&b[(size_t)t * Npad], rhs + (m0 + t) * N, (size_t)N * 8);
            // (no synchronisation after the upload: `b` lives until the download below has synchronised the stream)
            if (h->dtype == 64) {
                HIPCHK(hipMemcpyAsync(h->dTrsvX, b.data(), b.size() * 8, hipMemcpyHostToDevice, h->stream));
            } else {
                b32.assign(b.begin(), b.end());
                HIPCHK(hipMemcpyAsync(h->dTrsvX, b32.data(), b32.size() * 4, hipMemcpyHostToDevice, h->stream));
            }
            // forward: input -> pass 0; backward: pass 0 -> pass 1
            int rc;
            void* xres = nullptr;
            if (h->dtype == 64) {
                double *x0 = nullptr, *x1 = nullptr;
                rc = queue_trsv_fill<double>(h, nr, 2);
                if (!rc) rc = queue_trsv<double>(h, trsv_input<double>(h), 0, nr, false, &x0);
                if (!rc) rc = queue_trsv<double>(h, x0, 1, nr, true, &x1);
                xres = x1;
            } else {
                float *x0 = nullptr, *x1 = nullptr;
                rc = queue_trsv_fill<float>(h, nr, 2);
                if (!rc) rc = queue_trsv<float>(h, trsv_input<float>(h), 0, nr, false, &x0);
                if (!rc) rc = queue_trsv<float>(h, x0, 1, nr, true, &x1);
                xres = x1;
            }
            if (rc) return rc;
            rc = queue_abort_probe(h);
            if (rc) return rc;
            rc = DISPATCH(h, download, h, b, xres, (size_t)nr * Npad, h->stream);
            if (rc) return rc;
            HIPCHK(hipGetLastError());
            harvest(h);
            rc = abort_probe_verdict(h, "single-vector substitution timed out (set option trsv=0 and report)");
            if (rc) return rc;
            for (int t = 0; t < nr; ++t) memcpy(out + (m0 + t) * N, &b[(size_t)t * Npad], (size_t)N * 8);
        }
        return GPHIP_OK;
    }
    int rc = ensure_vbuf(h, nrhs < MC ? (nrhs + TB - 1) / TB * TB : MC);
    if (rc) return rc;
    h->cs = h->stream;
    std::vector<double> v;
    for (int64_t m0 = 0; m0 < nrhs; m0 += MC) {
        const int64_t mc = (nrhs - m0 < MC) ? (nrhs - m0) : MC;
        const int64_t mpad = (mc + TB - 1) / TB * TB;
        // few vectors in a 128-row block: move the vectors, not the zero padding (rows_to_vblock_kernel); the staging block is the
        // single-vector path's scratch when it exists and is large enough, else a buffer of its own
        const bool compact = mc * 4 <= mpad * 3;
        void* dRows = nullptr;
        if (compact) {
            const size_t need = (size_t)mc * Npad * h->es;
            if (need > h->rows_cap) {
                (void)hipFree(h->dRows);
                h->dRows = nullptr; h->rows_cap = 0;
                if (hipMalloc(&h->dRows, need) == hipSuccess) h->rows_cap = need;
                else (void)hipGetLastError();
            }
            dRows = h->rows_cap >= need ? h->dRows : nullptr;
        }
        if (dRows) {
            v.assign((size_t)mc * Npad, 0.0);
            for (int64_t t = 0; t < mc; ++t) memcpy(&v[(size_t)t * Npad], rhs + (m0 + t) * N, (size_t)N * 8);
            rc = DISPATCH(h, upload, h, dRows, v, h->stream);
            if (rc) return rc;
            const unsigned gx = (unsigned)((Npad + 255) / 256);
            if (h->dtype == 64) hipLaunchKernelGGL(rows_to_vblock_kernel<double>, dim3(gx), dim3(256), 0, h->stream, (const double*)dRows, (int)mc, (long)Npad, (double*)h->dV, (long)mpad);
            else hipLaunchKernelGGL(rows_to_vblock_kernel<float>, dim3(gx), dim3(256), 0, h->stream, (const float*)dRows, (int)mc, (long)Npad, (float*)h->dV, (long)mpad);
        } else {
            v.assign((size_t)mpad * Npad, 0.0);
            for (int64_t t = 0; t < mc; ++t)
                for (int64_t j = 0; j < N; ++j) v[(size_t)j * mpad + t] = rhs[(m0 + t) * N + j];
            rc = DISPATCH(h, upload, h, h->dV, v, h->stream);
            if (rc) return rc;
        }
        // after a single-launch fit both halves are ONE dataflow launch each (forward as in gphip_predict; backward over a copy of
        // the factor with its 64 x 64 blocks transposed, made on the first solve of a fit)
        ensure_w64(h);                         // (also after a look-ahead-schedule fit: 64-block inverses cut out of the 128-block ones)
        const bool dfs = df_forward_ok(h, mpad);
        if (dfs) launch_dataflow_inverse<double, 64>(h, mpad);
        else DISPATCH(h, queue_forward_rows, h, mpad, 1);
        if (dfs && df_backward_ready<double>(h)) launch_dataflow_inverse<double, 64>(h, mpad, true);
        else DISPATCH(h, queue_backward_rows, h, mpad);
        if (dfs) { rc = queue_abort_probe(h); if (rc) return rc; }
        if (dRows) {
            const unsigned gx = (unsigned)((Npad + 255) / 256);
            if (h->dtype == 64) hipLaunchKernelGGL(vblock_to_rows_kernel<double>, dim3(gx), dim3(256), 0, h->stream, (const double*)h->dV, (long)mpad, (int)mc, (long)Npad, (double*)dRows);
            else hipLaunchKernelGGL(vblock_to_rows_kernel<float>, dim3(gx), dim3(256), 0, h->stream, (const float*)h->dV, (long)mpad, (int)mc, (long)Npad, (float*)dRows);
            rc = DISPATCH(h, download, h, v, dRows, (size_t)mc * Npad, h->stream);
        } else {
            rc = DISPATCH(h, download, h, v, h->dV, (size_t)mpad * Npad, h->stream);
        }
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        harvest(h);
        if (dfs) { rc = abort_probe_verdict(h, "dataflow substitution timed out (set option predict_df=0 and report)"); if (rc) return rc; }
        if (dRows) {
            for (int64_t t = 0; t < mc; ++t) memcpy(out + (m0 + t) * N, &v[(size_t)t * Npad], (size_t)N * 8);
        } else {
            for (int64_t t = 0; t < mc; ++t)
                for (int64_t j = 0; j < N; ++j) out[(m0 + t) * N + j] = v[(size_t)j * mpad + t];
        }
    }
    return GPHIP_OK;
}

// ------------------------------------------------------------------------------------------
// Multi-GPU 1-D block-cyclic Cholesky: per-rank compute steps.  The host
// (bayesianinference_amd/dist_cholesky.py) owns the schedule and moves factored panels between
// ranks with torch.distributed broadcast (RCCL over xGMI); outer panel j (h->panel tile columns)
// belongs to rank j % world, the rhs x rhs corner tile to rank 0.
// ------------------------------------------------------------------------------------------
int gphip_set_streams(gphip_handle h, void* main_stream, void* panel_stream) {
    if (!h || !main_stream || !panel_stream) return fail(h, GPHIP_ERR_ARG, "null stream");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    if (h->own_streams) {
        HIPCHK(hipStreamSynchronize(h->pstream));
        HIPCHK(hipStreamSynchronize(h->stream));
        (void)hipStreamDestroy(h->pstream);
        (void)hipStreamDestroy(h->stream);
        h->own_streams = false;
    }
    h->stream = static_cast<hipStream_t>(main_stream);
    h->pstream = static_cast<hipStream_t>(panel_stream);
    h->cs = h->stream;
    return GPHIP_OK;
}

int gphip_dist_num_panels(gphip_handle h, int* nouter) {
    if (!h || !nouter) return GPHIP_ERR_ARG;
    *nouter = (int)((h->Nt + h->panel - 1) / h->panel);
    return GPHIP_OK;
}

// shape (in elements of the handle's dtype) of packed panel k: rows = all tile rows from the
// panel's first diagonal block down to and including the rhs block-row, cols = the panel's width
int gphip_dist_panel_shape(gphip_handle h, int k, int64_t* rows, int64_t* cols) {
    if (!h || !rows || !cols) return GPHIP_ERR_ARG;
    const int64_t K0 = (int64_t)k * h->panel, K1 = (K0 + h->panel < h->Nt) ? K0 + h->panel : h->Nt;
    if (k < 0 || K0 >= h->Nt) return fail(h, GPHIP_ERR_DIM, "panel index out of range");
    // a packed panel IS the panel's contiguous range of the tile-major workspace (tile columns K0 .. K1-1, each from its
    // diagonal tile down to the rhs tile row): rows x cols = its element count x 1, opaque to the host
    *rows = (tile_index((int)K1, (int)K1, (int)h->R) - tile_index((int)K0, (int)K0, (int)h->R)) * TS;
    *cols = 1;
    return GPHIP_OK;
}

int gphip_dist_begin(gphip_handle h, const double* theta, int p, int rank, int world) {
    if (!h || !theta) return fail(h, GPHIP_ERR_ARG, "null argument");
    if (p != h->p) return fail(h, GPHIP_ERR_DIM, "theta has the wrong length");
    if (world < 1 || rank < 0 || rank >= world) return fail(h, GPHIP_ERR_ARG, "bad rank/world");
    if (h->kernel_id == GPHIP_KERNEL_NULL) return fail(h, GPHIP_ERR_UNSUPPORTED, "null kernel needs no factorisation");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    const bool full = h->replicate_factor != 0;
    int rc = ensure_slots(h, 1, full);
    if (rc) return rc;
    rc = dist_layout(h, rank, world, full);
    if (rc) return rc;
    invalidate_fit(h);
    h->dist_rank = rank; h->dist_world = world;
    h->dist_first_factored = -1;
    h->dist_theta_ok = stage_theta(h, 0, theta, h->pw_nug_host, h->pw_mean_host);
    rc = upload_pw(h, 0, 1);                   // point-dependent nugget / mean of this evaluation, if the caller set them
    if (rc) return rc;
    rc = copy_theta(h, 1);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(h->dInfo, 0, 4, h->stream));
    // panel schedule of the owner (dist_panel_df; -1 = the library's choice).  Round 6, by scripts/scale_model.py on measured step
    // times: 3 (ONE dataflow launch per panel incl. the look-ahead update, every tile column handed to the broadcast stream by a
    // counter the launch bumps) from 2 ranks on; where the device has no stream-ordered wait on memory: 2 at two ranks, else 0.
    const bool df_able = h->dtype == 64 && h->dataflow != 0;
    h->dist_df_mode = !df_able ? 0 : h->dist_panel_df < 0 ? (world >= 2 ? 3 : 0) : h->dist_panel_df;
    if (h->dist_df_mode >= 3) {                // column signals: needs stream-ordered waits on device memory
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, h->device) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (can && !h->dColSig && hipMalloc((void**)&h->dColSig, 64 * sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); h->dColSig = nullptr; }
        if (!can || !h->dColSig || h->panel > 64) h->dist_df_mode = (h->dist_panel_df < 0 && world != 2) ? 0 : 2;
        else {
            HIPCHK(hipMemsetAsync(h->dColSig, 0, 64 * sizeof(unsigned int), h->pstream));
            for (unsigned int& t : h->colsig_target) t = 0;
        }
    }
    h->dist_df_active = h->dist_df_mode != 0;
    h->df_prev_ptr = nullptr; h->df_prev_k = -2;
    HIPCHK(hipMemsetAsync(h->dPartial, 0, (size_t)2 * h->Nt * 8, h->stream));
    h->cs = h->stream;
    DISPATCH(h, queue_build, h, 1);
    h->pw_mean_on = h->pw_nug_on = false;
    return GPHIP_OK;
}

// owner of panel k: factor it in place (panel stream).  `packed` (optional): a copy of the panel's range for a host that
// runs its own collective (dist_cholesky.py); the in-library schedule broadcasts straight from the rank's storage
int gphip_dist_factor_panel(gphip_handle h, int k, void* packed) {
    if (!h) return fail(h, GPHIP_ERR_ARG, "null argument");
    int64_t rows, cols;
    int rc = gphip_dist_panel_shape(h, k, &rows, &cols);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->dist_world < 1) return fail(h, GPHIP_ERR_STATE, "gphip_dist_factor_panel outside dist_begin/dist_end");
    if (k % h->dist_world != h->dist_rank) return fail(h, GPHIP_ERR_ARG, "gphip_dist_factor_panel: not the owner of this panel");
    HIPCHK(hipSetDevice(h->device));
    const int64_t K0 = (int64_t)k * h->panel;
    const int64_t K1 = (K0 + h->panel < h->Nt) ? K0 + h->panel : h->Nt;
    h->cs = h->pstream;
    h->ws_override = dist_panel_base(h, k);
    const bool first_factored = h->dist_first_factored == k;       // (by this rank's look-ahead update of the panel)
    h->dist_first_factored = -1;
    if (h->dist_df_active) {
        // Latency-shaped owner path (round 4): the whole panel -- its 2 nin 64-wide tile columns over all rows below -- as ONE
        // dataflow launch on the panel stream: the chain is 2 nin flag hops (~22 us each) instead of 3 dependent launches per
        // 128-wide column.  Meant for an owner whose chip is mostly idle (its share of the trailing update is 1 / world).
        // dist_panel_df = 2: the look-ahead update of this panel by panel k - 1 rides in the SAME launch
        // (the tasks read panel k - 1 from where the broadcast put it, as 2 P more slabs): the update's MFMA work then fills
        // the chip around the panel's serial chain instead of preceding it.
        hipStream_t keep = h->stream;
        h->stream = h->pstream;
        if (h->df_prev_ptr && h->df_prev_k == k - 1 && k >= 1) {
            const int64_t Kp = K0 - h->panel;
            const char* pbase = static_cast<const char*>(h->df_prev_ptr) - dist_panel_first(h, k - 1) * TS * (long)h->es;
            // (three workgroups per CU once the launch is throughput bound: the early, tall panels)
            unsigned int* sig = h->dist_df_mode >= 3 ? h->dColSig : nullptr;
            if (h->dataflow_occ3 > 0)
                launch_dataflow<double, 64, 3>(h, 1, 2 * (int)Kp, nullptr, 0, 2 * (int)(K1 - Kp), 2 * (int)(K0 - Kp), pbase, sig);
            else
                launch_dataflow<double, 64>(h, 1, 2 * (int)Kp, nullptr, 0, 2 * (int)(K1 - Kp), 2 * (int)(K0 - Kp), pbase, sig);
            if (sig) {
                // tiles the launch makes final per tile column: 64-columns j0, j0 + 1 of a submatrix with R64 tile rows (rhs row included)
                const int nprev = 2 * (int)(K0 - Kp), R64 = 2 * (int)(h->Nt - Kp) + 1;
                for (int c = 0; c < (int)(K1 - K0); ++c) {
                    const int j0 = nprev + 2 * c;
                    h->colsig_target[c] += (unsigned int)((R64 - j0) + (R64 - j0 - 1));
                    if (h->col_waits) h->col_waits->push_back({h->dColSig + c, h->colsig_target[c]});
                }
            }
        } else {
            unsigned int* sig = h->dist_df_mode >= 3 ? h->dColSig : nullptr;
            launch_dataflow<double, 64>(h, 1, 2 * (int)K0, nullptr, 0, 2 * (int)(K1 - K0), 0, nullptr, sig);
            if (sig) {
                const int R64 = 2 * (int)(h->Nt - K0) + 1;
                for (int c = 0; c < (int)(K1 - K0); ++c) {
                    h->colsig_target[c] += (unsigned int)((R64 - 2 * c) + (R64 - 2 * c - 1));
                    if (h->col_waits) h->col_waits->push_back({h->dColSig + c, h->colsig_target[c]});
                }
            }
        }
        h->df_prev_ptr = nullptr; h->df_prev_k = -2;
        h->stream = keep;
        if (h->dist_df_mode >= 3) {
            // a launch that never ran would leave the column counters short of their targets and the waits on them pending for
            // ever: report it NOW (the caller then skips the waits: drain mode)
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) { h->ws_override = nullptr; h->cs = h->stream; if (h->col_waits) h->col_waits->clear(); return fail(h, GPHIP_ERR_HIP, hipGetErrorString(le)); }
        }
    } else {
        DISPATCH(h, queue_panel, h, (int)K0, (int)(K1 - K0), 1, first_factored);
    }
    h->ws_override = nullptr;
    if (packed && packed != (void*)dist_panel_range(h, k))
        HIPCHK(hipMemcpyAsync(packed, dist_panel_range(h, k), (size_t)rows * cols * h->es, hipMemcpyDeviceToDevice, h->pstream));
    h->cs = h->stream;
    return GPHIP_OK;
}

// apply panel k (read from `packed`) to the outer panels j in [j_first, j_last) this rank owns;
// j == num_panels addresses the rhs x rhs corner tile (rank 0).  on_panel_stream selects the stream.
int gphip_dist_update(gphip_handle h, int k, const void* packed, int j_first, int j_last, int on_panel_stream) {
    if (!h || !packed) return fail(h, GPHIP_ERR_ARG, "null argument");
    int64_t rows, cols;
    int rc = gphip_dist_panel_shape(h, k, &rows, &cols);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->dist_world < 1) return fail(h, GPHIP_ERR_STATE, "gphip_dist_update outside dist_begin/dist_end");
    HIPCHK(hipSetDevice(h->device));
    const int Nt = (int)h->Nt, R = Nt + 1, P = h->panel;
    const int nouter = (Nt + P - 1) / P;
    const long K0 = (long)k * P;
    cols = (std::min<long>(K0 + P, Nt) - K0) * TB;         // contraction length = the panel's width in columns
    h->cs = on_panel_stream ? h->pstream : h->stream;
    // The owned panels j = j0, j0 + world, .. of this update go out as ONE grouped launch (blockIdx.y = owned panel;
    // gemm_nt decodes its own column range) instead of one launch per 512-column strip: a rank of an 8-GPU job owns
    // up to 8 strips per step, and a world of one owns them all (then they are adjacent: one plain triangular launch).
    const int W = h->dist_world, jb = (j_first > k + 1 ? j_first : k + 1), je = (j_last < nouter ? j_last : nouter);
    int j0 = -1, cnt = 0;
    for (int j = jb; j < je; ++j)
        if (j % W == h->dist_rank) { if (j0 < 0) j0 = j; ++cnt; }
    const int cls = on_panel_stream ? 3 : 4;
    const bool corner = j_last > nouter && nouter >= jb && h->dist_rank == 0;
    if (on_panel_stream && h->dist_df_mode >= 2 && cnt == 1 && j0 == k + 1 && je - jb == 1 && !corner) {
        // dist_panel_df = 2: the look-ahead update of panel k + 1 is DEFERRED into that panel's dataflow launch
        // (gphip_dist_factor_panel), which reads panel k from `packed` -- the caller keeps it alive until then, as both the
        // in-library schedule (three rotating receive buffers) and dist_cholesky.py do.
        h->df_prev_ptr = packed; h->df_prev_k = k;
        h->cs = h->stream;
        return GPHIP_OK;
    }
    // (dist_first_factored stays set until gphip_dist_factor_panel consumes it: the look-ahead update is the LAST update of
    //  its panel, so a main-stream update issued between LA(k) and the factorisation of panel k + 1 cannot invalidate it)
    if (cnt > 0) {
        // the look-ahead update of the NEXT panel (this rank owns it) also factors that panel's first diagonal block
        // (fuse_potrf): gphip_dist_factor_panel then starts at the panel solve
        const bool la_one = on_panel_stream && h->fuse_potrf && je - jb == 1 && !h->dist_df_active;
        if (W == 1) {            // adjacent panels (and the corner tile right behind them): one triangular launch
            if (la_one) h->fuse_b = j0 * P;
            DISPATCH(h, queue_dist_update, h, packed, K0, (long)rows, (long)cols, j0 * P,
                     (corner && je == nouter) ? R : std::min(je * P, Nt), cls);
            if (la_one && h->fuse_done) h->dist_first_factored = j0;
        } else if (cnt == 1) {
            if (la_one) h->fuse_b = j0 * P;
            DISPATCH(h, queue_dist_update, h, packed, K0, (long)rows, (long)cols, j0 * P, std::min(j0 * P + P, Nt), cls);
            if (la_one && h->fuse_done) h->dist_first_factored = j0;
        } else {
            DISPATCH(h, queue_dist_update, h, packed, K0, (long)rows, (long)cols, j0 * P, Nt, cls, cnt, W * P);
        }
    }
    if (corner && !(cnt > 0 && W == 1 && je == nouter))
        DISPATCH(h, queue_dist_update, h, packed, K0, (long)rows, (long)cols, Nt, R, cls);
    h->cs = h->stream;
    return GPHIP_OK;
}

// local results: logdet_partial = 2 * sum over owned diagonal blocks of sum(log L_ii); quad is
// meaningful on rank 0 only (0 elsewhere); the host all-reduces (sum, sum, max).
int gphip_dist_end(gphip_handle h, double* logdet_partial, double* quad, int* info) {
    if (!h || !logdet_partial || !quad || !info) return fail(h, GPHIP_ERR_ARG, "null argument");
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (h->dist_world < 1) return fail(h, GPHIP_ERR_STATE, "gphip_dist_end without gphip_dist_begin");
    HIPCHK(hipSetDevice(h->device));
    DISPATCH(h, queue_finalize, h);
    HIPCHK(hipMemcpyAsync(h->hRes, h->dRes, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->hInfo, h->dInfo, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->pstream));
    if (h->dist_df_active)                     // dataflow panels: did a dependency wait hit its spin limit?
        HIPCHK(hipMemcpyAsync(h->hInfo + 1, reinterpret_cast<int*>(h->dTicket + 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    harvest(h);
    if (h->dist_df_active && h->hInfo[1] != 0) {
        HIPCHK(hipMemsetAsync(h->dTicket + 1, 0, 8 + DF_PARK_SLOTS * 4, h->stream));
        h->dist_world = 0; h->dist_rank = 0;
        return fail(h, GPHIP_ERR_HIP, "dataflow panel schedule timed out (set option dist_panel_df=0 and report)");
    }
    *logdet_partial = h->hRes[0];
    *quad = (h->dist_rank == 0) ? h->hRes[1] : 0.0;
    *info = h->dist_theta_ok ? h->hInfo[0] : GPHIP_INFO_NAN;
    h->dist_world = 0; h->dist_rank = 0;
    return GPHIP_OK;
}

namespace {
// one table for set / get / the GPHIP_OPTIONS environment override
int* option_slot(gphip_ctx* h, const char* name) {
    struct Entry { const char* name; int gphip_ctx::*field; };
    static const Entry table[] = {
        {"panel", &gphip_ctx::panel}, {"profile", &gphip_ctx::profile}, {"xcd_swizzle", &gphip_ctx::swizzle},
        {"lookahead", &gphip_ctx::lookahead}, {"supertile", &gphip_ctx::supertile}, {"panel_wide", &gphip_ctx::panel_wide},
        {"panel_left", &gphip_ctx::panel_left}, {"thin_tiles", &gphip_ctx::thin_tiles}, {"fuse_potrf", &gphip_ctx::fuse_potrf},
        {"latency_gemm", &gphip_ctx::latency_gemm}, {"latency_max_nt", &gphip_ctx::latency_max_nt},
        {"dataflow", &gphip_ctx::dataflow}, {"dataflow_max_nt", &gphip_ctx::dataflow_max_nt},
        {"dataflow_max_slots", &gphip_ctx::dataflow_max_slots}, {"dataflow_max_tasks", &gphip_ctx::dataflow_max_tasks}, {"dataflow_fine_nt", &gphip_ctx::dataflow_fine_nt},
        {"dataflow_tail", &gphip_ctx::dataflow_tail}, {"dataflow_lds_kib", &gphip_ctx::dataflow_lds_kib},
        {"dataflow_occ3", &gphip_ctx::dataflow_occ3},
        {"fused_eval", &gphip_ctx::fuse_option}, {"panel_df", &gphip_ctx::panel_df}, {"grad_potri", &gphip_ctx::grad_potri}, {"predict_df", &gphip_ctx::predict_df},
        {"kbuild_mfma", &gphip_ctx::kbuild_mfma}, {"kbuild_mfma_bound", &gphip_ctx::kbuild_mfma_bound},
        {"kbuild_mfma_digits", &gphip_ctx::kbuild_mfma_digits}, {"trsv", &gphip_ctx::trsv}, {"predict_df_max_nt", &gphip_ctx::predict_df_max_nt}, {"last_issue_us", &gphip_ctx::last_issue_us}, {"last_dist_panel_df", &gphip_ctx::dist_df_mode},
        {"custom_grad", &gphip_ctx::custom_grad}, {"grad_analytic", &gphip_ctx::grad_analytic},
        {"max_slots", &gphip_ctx::max_slots}, {"shard_min_n", &gphip_ctx::shard_min_n},
        {"replicate_factor", &gphip_ctx::replicate_factor}, {"share_local_panels", &gphip_ctx::share_local_panels},
        {"bcast_chunks", &gphip_ctx::bcast_chunks}, {"bcast_two_hop", &gphip_ctx::bcast_two_hop}, {"dist_panel_df", &gphip_ctx::dist_panel_df}, {"dist_owner_yield", &gphip_ctx::dist_owner_yield},
        {"debug_fail_alloc", &gphip_ctx::debug_fail_alloc}, {"debug_fail_hip", &gphip_ctx::debug_fail_hip},
    };
    // fault injection ("debug_*") exists for the test-suite only: the names resolve in a process that was started with
    // GPHIP_TEST_HOOKS=1 and nowhere else (not through GPHIP_OPTIONS either: apply_env_options skips them)
    static const bool test_hooks = [] { const char* e = getenv("GPHIP_TEST_HOOKS"); return e && !strcmp(e, "1"); }();
    if (!strncmp(name, "debug_", 6) && !test_hooks) return nullptr;
    for (const Entry& e : table)
        if (!strcmp(name, e.name)) return &(h->*(e.field));
    return nullptr;
}

// GPHIP_OPTIONS="dataflow=0,panel=6": defaults for every handle of the process (hosts that cannot call
// gphip_set_option, e.g. through the LibraryLink shim; debugging).  Unknown names / bad values are ignored.
void apply_env_options(gphip_ctx* h) {
    const char* env = getenv("GPHIP_OPTIONS");
    if (!env) return;
    std::string str(env);
    size_t pos = 0;
    while (pos < str.size()) {
        size_t end = str.find(',', pos);
        if (end == std::string::npos) end = str.size();
        const std::string item = str.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) {
            std::string key = item.substr(0, eq);
            key.erase(0, key.find_first_not_of(" \t"));
            key.erase(key.find_last_not_of(" \t") + 1);
            char* stop = nullptr;
            const double v = strtod(item.c_str() + eq + 1, &stop);
            if (stop != item.c_str() + eq + 1 && key.compare(0, 6, "debug_") != 0) (void)gphip_set_option(h, key.c_str(), v);
        }
        pos = end + 1;
    }
    h->err.clear();
}
}  // namespace

int gphip_set_option(gphip_handle h, const char* name, double value) {
    if (!h || !name) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    const int v = (int)value;
    int* slot = option_slot(h, name);
    if (!slot) return fail(h, GPHIP_ERR_ARG, "unknown option");
    if (!strcmp(name, "panel") && (v < 1 || v > 64)) return fail(h, GPHIP_ERR_ARG, "panel out of range");
    if (!strcmp(name, "max_slots") && v < 1) return fail(h, GPHIP_ERR_ARG, "max_slots < 1");
    *slot = v;
    if (h->group)                              // every member of a multi-device handle runs the same schedule
        for (size_t i = 1; i < h->group->members.size(); ++i) (void)gphip_set_option(h->group->members[i], name, value);
    return GPHIP_OK;
}

int gphip_get_option(gphip_handle h, const char* name, double* value) {
    if (!h || !name || !value) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    const int* slot = option_slot(h, name);
    if (!slot) return fail(h, GPHIP_ERR_ARG, "unknown option");
    *value = (double)*slot;
    return GPHIP_OK;
}

int gphip_get_profile(gphip_handle h, int cls, double* ms, double* launches, double* flops, double* bytes) {
    if (!h || cls < 0 || cls >= GPHIP_NCLASS) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    if (ms) *ms = h->acc_ms[cls];
    if (launches) *launches = h->acc_n[cls];
    if (flops) *flops = h->acc_flops[cls];
    if (bytes) *bytes = h->acc_bytes[cls];
    return GPHIP_OK;
}

int gphip_reset_profile(gphip_handle h) {
    if (!h) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    for (int c = 0; c < GPHIP_NCLASS; ++c) h->acc_ms[c] = h->acc_n[c] = h->acc_flops[c] = h->acc_bytes[c] = 0;
    return GPHIP_OK;
}

// Device memory this handle's rank `member` (0 .. nlocal-1; 0 for a plain handle) holds for factor storage right now:
// the dense workspace slots + its compact own-panel storage + its receive buffers (diagnostics / tests).
int gphip_factor_bytes(gphip_handle h, int member, double* bytes) {
    if (!h || !bytes) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    gphip_ctx* m = h;
    if (h->group) {
        if (member < 0 || member >= (int)h->group->members.size()) return fail(h, GPHIP_ERR_ARG, "no such local rank");
        m = h->group->members[(size_t)member];
    } else if (member != 0) return fail(h, GPHIP_ERR_ARG, "no such local rank");
    double b = (m->dA ? (double)m->slots * (double)m->slot_elems * (double)m->es : 0.0) + (double)m->own_bytes;
    for (void* pk : m->packed) b += pk ? (double)m->packed_bytes : 0.0;
    *bytes = b;
    return GPHIP_OK;
}

int gphip_sync(gphip_handle h) {
    if (!h) return GPHIP_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(h->mu);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->pstream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GPHIP_OK;
}

}  // extern "C"

#include "gphip_hostlogic.inc"
#include "gphip_sampler.inc"
