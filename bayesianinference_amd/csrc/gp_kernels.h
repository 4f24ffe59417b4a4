// gp_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the GP likelihood path.
//
//   k_scale        xs = x / l                       (prologue of K1)
//   kbuild_kernel  K1/K7: pairwise kernel matrix tiles, lower triangle, straight into the
//                  Cholesky workspace (BGP:29-43 covarianceMatrix; BGP:100-109 cross form)
//   potrf128       K2a + K3: 128x128 diagonal-block Cholesky + triangular inverse in LDS (MFMA),
//                  log-det partial, SPD test
//   gemm_nt        K2b/K2c on fp64 MFMA (v_mfma_f64_16x16x4_f64): mode 1 = panel solve X <- X W^T
//                  (also carries r -> z = L^-1 r, K4), mode 0 = C -= A B^T SYRK/GEMM trailing update
//   finalize       log det = 2 sum log L_ii, quad = |z|^2, info
//   predict_reduce K9/K10 epilogue: mu* and var* from V = k*^T L^-T and z
//
// Storage: one column-major workspace matrix per batch slot, leading dimension ld = Npad + 128
// (Npad = N rounded up to the 128 tile).  Rows [Npad, Npad+128) carry right-hand sides as extra
// ROWS (row Npad = r^T): the panel solve and the trailing update then produce z^T = (L^-1 r)^T in
// that row and -|z|^2 in element (Npad, Npad) with no extra kernels ("bordered" Cholesky).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gphip {

constexpr int TB = 128;         // tile edge
constexpr int SLOTP = 8;        // doubles of per-slot scalars: sf2, sn2, mu, pivot_tol, bad_theta

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// ---------------------------------------------------------------------------------------------
// exp(x) for x <= 0, fp64, no hardware transcendental on gfx950: Cody-Waite reduction by ln2,
// degree-13 Taylor/Horner on |r| <= ln2/2 (truncation 4e-18), v_ldexp_f64 scaling. ~20 VALU ops.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -800.0);
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;              // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868099e-09);   // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441719e-08);   // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985891e-07);   // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587302e-05);   // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841e-04);   // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889e-03);   // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332e-03);   // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// KT = 0: squared exponential  sf2 * exp(-r2/2)
// KT = 1: Matern-5/2           sf2 * (1 + s5 + 5 r2/3) exp(-s5),  s5 = sqrt(5 r2)
template <int KT>
__device__ __forceinline__ double kfun(double r2, double sf2) {
    if (KT == 0) {
        return sf2 * exp_nonpos(-0.5 * r2);
    } else {
        const double s5 = __builtin_sqrt(5.0 * r2);
        return sf2 * (1.0 + s5 + (5.0 / 3.0) * r2) * exp_nonpos(-s5);
    }
}

// xs[slot][dd][i] = X[dd][i] * inv_ell[slot][dd]      (Xt is the transposed copy [d][npad])
__global__ void k_scale(const double* __restrict__ Xt, double* __restrict__ xs,
                        const double* __restrict__ inv_ell, int d, int npad) {
    const int slot = blockIdx.y;
    const long total = (long)d * npad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int dd = (int)(idx / npad);
        xs[(long)slot * total + idx] = Xt[idx] * inv_ell[slot * d + dd];
    }
}

// Decode linear index t of the lower triangle (incl. diagonal) of an n x n tile grid, enumerated
// column by column: column c holds rows c..n-1.
__device__ __forceinline__ void tri_decode(int t, int n, int& ti, int& tj) {
    const double b = 2.0 * n + 1.0;
    int c = (int)((b - __builtin_sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (c < 0) c = 0;
    if (c > n - 1) c = n - 1;
    // offset(c) = c*n - c(c-1)/2
    while (c > 0 && (long)c * n - (long)c * (c - 1) / 2 > t) --c;
    while ((long)(c + 1) * n - (long)(c + 1) * c / 2 <= t) ++c;
    tj = c;
    ti = c + (t - (int)((long)c * n - (long)c * (c - 1) / 2));
}

struct KBuildArgs {
    double* out;            // workspace base (slot 0)
    long ld;                // leading dimension (doubles)
    long bstride;           // doubles between slots
    const double* xi;       // scaled I-operand points [slot][D][npad_i]  (rows of the output)
    const double* xj;       // scaled J-operand points [slot][D][npad_j]  (columns of the output)
    long xi_bstride, xj_bstride;
    int npad_i, npad_j;     // padded point counts (multiples of 128)
    int n_i, n_j;           // true point counts
    const double* y;        // [npad_j] outputs (mode 0 only)
    const double* slotp;    // [slot][SLOTP]
    int d;                  // runtime dimension (used when D == 0)
    int mode;               // 0: train x train (lower-tri tiles, nugget, identity padding, rhs rows)
                            // 1: cross  (rectangular tiles, zero padding)
    int nt_i, nt_j;         // tile counts; mode 0: nt_i = nt_j + 1 (extra rhs block-row)
    int own_panel, own_world, own_rank;   // own_world > 0 (multi-GPU 1-D block-cyclic layout): build
                            // only tile columns whose outer panel (tj / own_panel) belongs to own_rank;
                            // the rhs x rhs corner tile belongs to rank 0
};

// One 128x128 tile per workgroup (4 waves).  Wave w owns 32 output columns; lane owns 2 adjacent
// rows, so every store is a 16-byte dwordx4 and a wave writes one full 1 KiB column segment.
// The J-side points come from LDS as wave-uniform broadcasts; the I-side points live in registers.
template <int D, int KT>
__global__ __launch_bounds__(256) void kbuild_kernel(KBuildArgs a) {
    extern __shared__ double lds[];           // xj tile [d][128] (+ xi tile [d][128] when D == 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.y;
    int ti, tj;
    if (a.mode == 0) {
        tri_decode(blockIdx.x, a.nt_i, ti, tj);
    } else {
        ti = blockIdx.x % a.nt_i;
        tj = blockIdx.x / a.nt_i;
    }
    if (a.own_world > 0) {
        const int owner = (tj == a.nt_j) ? 0 : (tj / a.own_panel) % a.own_world;
        if (owner != a.own_rank) return;
    }
    const double* sp = a.slotp + (long)slot * SLOTP;
    const double sf2 = sp[0], sn2 = sp[1], mu = sp[2];
    double* out = a.out + (long)slot * a.bstride + (long)tj * TB * a.ld + (long)ti * TB;
    const int r0 = 2 * lane;

    if (a.mode == 0 && ti == a.nt_i - 1) {      // right-hand-side block-row: row 0 = r^T, rest 0
        for (int jj = wave * 32; jj < wave * 32 + 32; ++jj) {
            const int gj = tj * TB + jj;
            double2 v = make_double2(0.0, 0.0);
            if (lane == 0 && tj < a.nt_j && gj < a.n_j) v.x = a.y[gj] - mu;
            *reinterpret_cast<double2*>(out + (long)jj * a.ld + r0) = v;
        }
        return;
    }

    const int d = (D > 0) ? D : a.d;
    const double* xjg = a.xj + (long)slot * a.xj_bstride + (long)tj * TB;
    const double* xig = a.xi + (long)slot * a.xi_bstride + (long)ti * TB;
    double* xjs = lds;
    double* xis = lds + d * TB;
    for (int idx = tid; idx < d * TB; idx += 256) {
        const int dd = idx >> 7, c = idx & 127;
        xjs[idx] = xjg[(long)dd * a.npad_j + c];
        if (D == 0) xis[idx] = xig[(long)dd * a.npad_i + c];
    }
    double xa[D > 0 ? D : 1], xb[D > 0 ? D : 1];
    if (D > 0) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const double2 v = *reinterpret_cast<const double2*>(xig + (long)dd * a.npad_i + r0);
            xa[dd] = v.x;
            xb[dd] = v.y;
        }
    }
    __syncthreads();

    const int gi = ti * TB + r0;
    const bool edge = (a.mode == 0) ? (ti == tj || (ti + 1) * TB > a.n_i)
                                    : ((ti + 1) * TB > a.n_i || (tj + 1) * TB > a.n_j);
    for (int jj = wave * 32; jj < wave * 32 + 32; ++jj) {
        double ra = 0.0, rb = 0.0;
        if (D > 0) {
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const double xjv = xjs[dd * TB + jj];
                const double da = xa[dd] - xjv, db = xb[dd] - xjv;
                ra = __builtin_fma(da, da, ra);
                rb = __builtin_fma(db, db, rb);
            }
        } else {
            for (int dd = 0; dd < d; ++dd) {
                const double xjv = xjs[dd * TB + jj];
                const double da = xis[dd * TB + r0] - xjv, db = xis[dd * TB + r0 + 1] - xjv;
                ra = __builtin_fma(da, da, ra);
                rb = __builtin_fma(db, db, rb);
            }
        }
        double va = kfun<KT>(ra, sf2), vb = kfun<KT>(rb, sf2);
        if (edge) {
            const int gj = tj * TB + jj;
            if (a.mode == 0) {
                if (gi == gj) va += sn2;
                if (gi + 1 == gj) vb += sn2;
                if (gj >= a.n_j || gi >= a.n_i) va = (gi == gj) ? 1.0 : 0.0;       // identity pad
                if (gj >= a.n_j || gi + 1 >= a.n_i) vb = (gi + 1 == gj) ? 1.0 : 0.0;
            } else {
                if (gj >= a.n_j || gi >= a.n_i) va = 0.0;
                if (gj >= a.n_j || gi + 1 >= a.n_i) vb = 0.0;
            }
        }
        *reinterpret_cast<double2*>(out + (long)jj * a.ld + r0) = make_double2(va, vb);
    }
}

// ---------------------------------------------------------------------------------------------
// potrf128: Cholesky of one 128x128 diagonal block AND its triangular inverse, in LDS.
//
// Factor phase, 8 panels of 16 columns:
//   (a) wave 0 factors the 16x16 diagonal block in registers (lane = row, v_readlane broadcasts),
//   (b) one thread per row below solves its 16 panel entries against L_pp,
//   (c) all 4 waves apply the rank-16 trailing update on v_mfma_f64_16x16x4_f64.
// Then L goes back to HBM, and W = L^-1 is formed in place (LAPACK dtrtri order, 16x16 blocks,
// MFMA products) and stored to the Winv workspace: the panel solve below the block is then a
// plain MFMA GEMM  X <- A W^T  (gemm_nt mode 1).
// SPD verdict: pivot <= tol (tol = 64 eps (sf2+sn2)) or NaN -> info = NOT_SPD (stands for
// LinearSolve::sing1/::luc -> Throw "MatInv", BGP:131-135).
// ---------------------------------------------------------------------------------------------
// LDS image: the 36 lower-triangle 16x16 tiles, tile (bi,bj) at ((bi(bi+1)/2 + bj) * 256) doubles,
// column-major inside the tile.  72 KiB: the kernel fits on a CU next to one gemm_nt workgroup
// (look-ahead runs it concurrently with the trailing SYRK), and MFMA fragment reads
// (lane -> row l&15, k = l>>4) touch 64 distinct banks.
constexpr int PT_LDS_DOUBLES = 36 * 256 + TB + 2;

__device__ __forceinline__ int ptile(int bi, int bj) { return ((bi * (bi + 1) / 2) + bj) << 8; }

__device__ __forceinline__ double readlane_d(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void potrf128_kernel(double* __restrict__ Abase, long ld,
                                                       long bstride, int b,
                                                       double* __restrict__ Winv, double* __restrict__ partial,
                                                       int nt, int* __restrict__ info,
                                                       const double* __restrict__ slotp) {
    extern __shared__ double Ls[];            // 36 tiles + dinv[128] + red[2]
    double* dinv = Ls + 36 * 256;
    double* red = dinv + TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int er = tid & 15, ec = tid >> 4;   // element (row, col) of a 16x16 tile owned in copies
    const int slot = blockIdx.x;
    double* Ad = Abase + (long)slot * bstride + (long)b * TB * (ld + 1);
    for (int bi = 0; bi < 8; ++bi)
        for (int bj = 0; bj <= bi; ++bj)
            Ls[ptile(bi, bj) + tid] = Ad[(long)(bj * 16 + ec) * ld + bi * 16 + er];
    const double tol = slotp[(long)slot * SLOTP + 3];
    bool bad = false;
    __syncthreads();

    // ------------------------------ factor phase ------------------------------
    for (int p = 0; p < 8; ++p) {
        double* Dpp = Ls + ptile(p, p);
        if (wave == 0) {                      // (a) 16x16 diagonal block, lane l15 owns row l15
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = Dpp[c * 16 + l15];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double dj = readlane_d(a[j], j);
                if (!(dj > tol)) { bad = true; dj = 1.0; }
                const double l = __builtin_sqrt(dj);
                const double rs = 1.0 / l;
                a[j] = (l15 == j) ? l : a[j] * rs;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    const double sc = readlane_d(a[j], c);
                    a[c] = __builtin_fma(-a[j], sc, a[c]);
                }
                if (lane == 0) dinv[16 * p + j] = rs;
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c <= l15) Dpp[c * 16 + l15] = a[c];
            }
        }
        __syncthreads();
        if (p == 7) break;
        if (tid < TB - 16 * p - 16) {         // (b) rows below: x L_pp^T = a, one row per thread
            double* Xr = Ls + ptile(p + 1 + (tid >> 4), p) + (tid & 15);
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = Xr[c * 16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double sacc = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) sacc = __builtin_fma(-x[k], Dpp[k * 16 + c], sacc);
                x[c] = sacc * dinv[16 * p + c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) Xr[c * 16] = x[c];
        }
        __syncthreads();
        {                                      // (c) trailing update C -= X X^T on MFMA
            const int t = 7 - p, ntile = t * (t + 1) / 2;
            for (int tt = wave; tt < ntile; tt += 4) {
                int u = 0;
                while ((u + 1) * (u + 2) / 2 <= tt) ++u;
                const int v = tt - u * (u + 1) / 2;
                const double* Xc = Ls + ptile(p + 1 + v, p) + l4 * 16 + l15;
                const double* Xrw = Ls + ptile(p + 1 + u, p) + l4 * 16 + l15;
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xc[kk * 64], Xrw[kk * 64], acc, 0, 0, 0);
                double* Ct = Ls + ptile(p + 1 + u, p + 1 + v) + l4 * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) Ct[r * 64] -= acc[r];
            }
        }
        __syncthreads();
    }

    // L back to HBM (lower-triangle tiles; diagonal tiles whole, their upper part is never read)
    for (int bi = 0; bi < 8; ++bi)
        for (int bj = 0; bj <= bi; ++bj)
            Ad[(long)(bj * 16 + ec) * ld + bi * 16 + er] = Ls[ptile(bi, bj) + tid];
    {   // sum log L_jj = -sum log dinv_j
        double lg = 0.0;
        if (tid < TB) lg = -log(dinv[tid]);
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
        if (tid < TB && lane == 0) red[wave] = lg;
    }

    // ------------------------------ inverse phase ------------------------------
    {   // (i) the eight 16x16 diagonal inverses: thread = (block, column)
        double w[16];
        const int blk = tid >> 4, c = tid & 15;
        double* Dbb = Ls + ptile(blk & 7, blk & 7);
        if (tid < TB) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double sacc = 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) sacc = __builtin_fma(Dbb[k * 16 + i], w[k], sacc);
                const double di = dinv[16 * blk + i];
                w[i] = (i < c) ? 0.0 : ((i == c) ? di : -sacc * di);
            }
        }
        __syncthreads();
        if (tid < TB) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i >= c) Dbb[c * 16 + i] = w[i];
        }
        __syncthreads();
    }
    for (int pb = 6; pb >= 0; --pb) {          // (ii) block column pb, rows q = pb+1..7
        const int t = 7 - pb;
        const double* Wpp = Ls + ptile(pb, pb);
        // step 1: T'_r = T_r W_pp  (W_pp lower triangular: mask k < j)
        d4 t1[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            t1[s] = (d4){0.0, 0.0, 0.0, 0.0};
            const int rr = wave + 4 * s;
            if (rr < t) {
                const double* Tr = Ls + ptile(pb + 1 + rr, pb);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 4 * kk + l4;
                    const double fa = Tr[k * 16 + l15];                               // T_r(i=l15, k)
                    const double fb = (k >= l15) ? Wpp[l15 * 16 + k] : 0.0;           // W_pp(k, j=l15)
                    t1[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, t1[s], 0, 0, 0);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = wave + 4 * s;
            if (rr < t) {
                double* Tr = Ls + ptile(pb + 1 + rr, pb);
#pragma unroll
                for (int r = 0; r < 4; ++r) Tr[l15 * 16 + l4 + 4 * r] = t1[s][r];
            }
        }
        __syncthreads();
        // step 2: W_q,pb = - sum_{r=pb+1..q} W_qr T'_r   (W_qq lower triangular: mask k > i)
        d4 t2[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            t2[s] = (d4){0.0, 0.0, 0.0, 0.0};
            const int qq = wave + 4 * s;
            if (qq < t) {
                const int q = pb + 1 + qq;
                for (int r = pb + 1; r <= q; ++r) {
                    const double* Wqr = Ls + ptile(q, r);
                    const double* Tr = Ls + ptile(r, pb);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = 4 * kk + l4;
                        double fa = Wqr[k * 16 + l15];                                // W_qr(i=l15, k)
                        if (r == q && k > l15) fa = 0.0;
                        const double fb = Tr[l15 * 16 + k];                           // T'_r(k, j=l15)
                        t2[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, t2[s], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int qq = wave + 4 * s;
            if (qq < t) {
                double* Wq = Ls + ptile(pb + 1 + qq, pb);
#pragma unroll
                for (int r = 0; r < 4; ++r) Wq[l15 * 16 + l4 + 4 * r] = -t2[s][r];
            }
        }
        __syncthreads();
    }
    // W to the workspace, dense column-major 128x128 with an explicit zero upper triangle
    double* Wg = Winv + ((long)slot * nt + b) * TB * TB;
    for (int bi = 0; bi < 8; ++bi)
        for (int bj = 0; bj < 8; ++bj) {
            double v = 0.0;
            if (bi > bj || (bi == bj && er >= ec)) v = Ls[ptile(bi, bj) + tid];
            Wg[(bj * 16 + ec) * TB + bi * 16 + er] = v;
        }
    if (tid == 0) {
        partial[(long)slot * nt + b] = red[0] + red[1];
        if (bad) info[slot] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_nt: C(i,j) -= sum_k A(i,k) B(j,k) for 128x128 tiles on v_mfma_f64_16x16x4_f64.
//
// 256 threads = 4 waves in a 2(i) x 2(j) arrangement, 64x64 per wave = 4x4 MFMA tiles, 64 fp64
// accumulators (128 VGPRs) per lane.  K is consumed in stages of 16 through double-buffered LDS
// (LDS-DMA global_load_lds_dwordx4 issued a stage ahead).  Both operand tiles are stored [k][row]
// with a padded leading dimension of 144 doubles so the MFMA fragment read
// (lane -> row = lane&15, k = lane>>4) is ds_read_b64 bank-conflict free.
//
// MFMA operand roles: the C-row (memory-contiguous) index i feeds the MFMA *B* operand so that
// D's column index (= lane&15) runs along contiguous memory of column-major C; the C-column index
// j feeds the *A* operand (D row = (lane>>4) + 4*reg; f64 layout, cdna_hip_programming.md §3).
// ---------------------------------------------------------------------------------------------
constexpr int GK = 16;          // K per LDS stage
constexpr int LDT = 144;        // padded LDS leading dimension (doubles)

struct GemmArgs {
    double* C; long ldc; long c_bstride;
    const double* A; long lda; long a_bstride;   // I operand: A(i,k) at A[i + k*lda]
    const double* B; long ldb; long b_bstride;   // J operand: B(j,k) at B[j + k*ldb]
    int K;                                       // multiple of 16
    int r0, r1, c0, c1;                          // tile ranges: rows [r0,r1), cols [c0,c1)
    int tri;                                     // 1: keep only tiles with ti >= tj (needs r0 >= c0)
    int nrect;                                   // tiles in the full-height rectangle part
    int ntiles;
    int swizzle;                                 // XCD-aware block remap
    int super;                                   // 1: pure-triangle launch enumerated in 8x8 super-tiles,
                                                 //    one super-tile per XCD at a time (L2 reuse)
    int mode;                                    // 0: C -= A B^T ; 1: C = A B^T (in-place panel solve
                                                 //    X <- X W^T: A aliases C, one column tile)
};

__device__ __forceinline__ void gemm_tile_decode(const GemmArgs& g, int t, int& ti, int& tj) {
    const int H = g.r1 - g.r0;
    if (!g.tri || t < g.nrect) {
        tj = g.c0 + t / H;
        ti = g.r0 + t % H;
    } else {
        int u, v;
        tri_decode(t - g.nrect, H, u, v);      // u >= v in an H x H triangle anchored at (r0, r0)
        ti = g.r0 + u;
        tj = g.r0 + v;
    }
}

// ROLE only gives each use its own kernel symbol (separate rows in rocprof summaries):
// 0 = trailing SYRK (K = panel*128, the dominant kernel), 1 = in-panel GEMM (K = 128),
// 2 = panel solve X <- X W^T (mode 1),
// 3 = "NN" form for the backward solve: the J operand is read transposed, B(j,k) at
//     B[k + j*ldb] (k contiguous), through an XOR-swizzled [j][16] LDS image; g.mode picks
//     C -= A B (0) or C = A B (1).
template <int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
    extern __shared__ double smem[];           // [2 stages][I: GK*LDT | J: GK*LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int slot = blockIdx.y;
    int bid = blockIdx.x;
    int ti, tj;
    if (g.super) {
        // Blocks b, b+8, b+16, .. run on XCD b % 8 (observed dispatch rule; speed only).  Give each
        // XCD whole 8x8 super-tiles: its 64 resident workgroups then share 8 row panels and 8
        // column panels through that XCD's private 4 MiB L2 instead of streaming 65 panels.
        const int H = g.r1 - g.r0, S = (H + 7) >> 3;
        const int x = bid & 7, q = bid >> 3;
        const int sidx = (q >> 6) * 8 + x, within = q & 63;
        if (sidx >= S * (S + 1) / 2) return;
        int I, J;
        tri_decode(sidx, S, I, J);
        const int u = 8 * I + (within & 7), v = 8 * J + (within >> 3);
        if (u >= H || v > u) return;
        ti = g.r0 + u;
        tj = g.r0 + v;
    } else {
        if (g.swizzle) {                       // give each XCD a contiguous chunk of the tile list
            const int nx = 8, n = g.ntiles;
            const int q = n / nx, rem = n % nx, x = bid % nx, o = bid / nx;
            bid = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + o;
        }
        gemm_tile_decode(g, bid, ti, tj);
    }

    const double* Ag = g.A + (long)slot * g.a_bstride + (long)ti * TB;
    const double* Bg = g.B + (long)slot * g.b_bstride + (ROLE == 3 ? (long)tj * TB * g.ldb : (long)tj * TB);
    // Staging: LDS-DMA (global_load_lds_dwordx4), no staging registers and no ds_write pass.
    // One wave-instruction moves one k-column of a tile: 64 lanes x 16 B = 128 rows = 1 KiB,
    // landing lane-linear at a wave-uniform LDS base (column kk at kk*LDT doubles, so the
    // padding sits between instructions).  Wave w moves columns w, w+4, w+8, w+12 of both tiles.
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const int srow = 2 * lane;
    auto stage = [&](int kb, int st) {
        double* Is = smem + st * (2 * GK * LDT);
        double* Js = Is + GK * LDT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = uw + 4 * s;
            const long kcol = (long)kb * GK + kk;
            __builtin_amdgcn_global_load_lds((glb_void*)(Ag + kcol * g.lda + srow),
                                             (lds_void*)(Is + kk * LDT), 16, 0, 0);
            if (ROLE != 3) {
                __builtin_amdgcn_global_load_lds((glb_void*)(Bg + kcol * g.ldb + srow),
                                                 (lds_void*)(Js + kk * LDT), 16, 0, 0);
            } else {
                // transposed source: instruction kk covers j = 8kk..8kk+7, lane -> (j, k-pair);
                // image Js[j*16 + 2*((k>>1) ^ (j&7)) + (k&1)] (swizzle applied on the source side)
                const int j = 8 * kk + (lane >> 3), kp = (lane & 7) ^ (j & 7);
                __builtin_amdgcn_global_load_lds((glb_void*)(Bg + (long)kb * GK + 2 * kp + (long)j * g.ldb),
                                                 (lds_void*)(Js + kk * 128), 16, 0, 0);
            }
        }
    };

    // C tile: lane holds i = i0 + y*16 + (lane&15), j = j0 + x*16 + (lane>>4) + 4r (f64 MFMA D layout)
    double* Cg = g.C + (long)slot * g.c_bstride + ((long)tj * TB + wj * 64 + (lane >> 4)) * g.ldc +
                 (long)ti * TB + wi * 64 + (lane & 15);
    const int nk = g.K / GK;
    stage(0, 0);
    // Update roles start the accumulators AT C (loads fly with the first DMA stage) and feed the
    // MFMA the negated J fragment, so acc ends as C - A B^T and the epilogue is stores only.
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            if (ROLE == 2 || (ROLE == 3 && g.mode == 1)) {
                acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[x][y][r] = Cg[(long)(x * 16 + 4 * r) * g.ldc + y * 16];
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int foff = (lane >> 4) * LDT + (lane & 15);
    for (int kb = 0; kb < nk; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nk) stage(kb + 1, cur ^ 1);       // DMA of the next stage flies under the MFMAs
        const double* Is = smem + cur * (2 * GK * LDT) + wi * 64 + foff;
        const double* Js = smem + cur * (2 * GK * LDT) + GK * LDT + wj * 64 + foff;
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            double fi[4], fj[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                fi[f] = Is[kk * 4 * LDT + f * 16];
                if (ROLE == 3) {
                    const int jrow = wj * 64 + f * 16 + (lane & 15), k = 4 * kk + (lane >> 4);
                    const double v = smem[cur * (2 * GK * LDT) + GK * LDT + jrow * 16 +
                                          2 * ((k >> 1) ^ (jrow & 7)) + (k & 1)];
                    fj[f] = (g.mode == 1) ? v : -v;
                } else {
                    fj[f] = (ROLE == 2) ? Js[kk * 4 * LDT + f * 16] : -Js[kk * 4 * LDT + f * 16];
                }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[x], fi[y], acc[x][y], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: stores only
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* cp = Cg + (long)(x * 16 + 4 * r) * g.ldc;
#pragma unroll
            for (int y = 0; y < 4; ++y) cp[y * 16] = acc[x][y][r];
        }
}

// log det = 2 sum partial ; quad = -E(0,0) ; res[slot] = {logdet, quad}
__global__ void finalize_kernel(const double* __restrict__ Abase, long ld, long bstride, int npad,
                                const double* __restrict__ partial, int nt, double* __restrict__ res) {
    const int slot = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nt; b += 64) s += partial[(long)slot * nt + b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (threadIdx.x == 0) {
        res[slot * 2 + 0] = 2.0 * s;
        res[slot * 2 + 1] = -Abase[(long)slot * bstride + (long)npad * ld + npad];
    }
}

// V: column-major [mpad x npad] (ld = mpad), V(t, j) = (L^-1 k*_t)_j.  z: row npad of the factor.
// mean[t] = mu + sum_j V(t,j) z_j ; var[t] = kappa - sum_j V(t,j)^2
__global__ void predict_reduce_kernel(const double* __restrict__ V, long ldv, int n,
                                      const double* __restrict__ zrow, long ldz, double mu,
                                      double kappa, int m, double* __restrict__ mean,
                                      double* __restrict__ var) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    double dot = 0.0, nrm = 0.0;
    for (int j = 0; j < n; ++j) {
        const double v = V[(long)j * ldv + t];
        dot = __builtin_fma(v, zrow[(long)j * ldz], dot);
        nrm = __builtin_fma(v, v, nrm);
    }
    mean[t] = mu + dot;
    var[t] = kappa - nrm;
}

}  // namespace gphip
