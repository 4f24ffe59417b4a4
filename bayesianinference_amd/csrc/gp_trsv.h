// gp_trsv.h -- triangular solves with ONE .. FOUR right-hand sides against the blocked factor, streaming L once per solve.
//
// The reference's "Inverse"[vector] (BGP:194, 407-412) and alpha = K^-1 r are two substitutions with a matrix that is read
// exactly once each: HBM-bound, 2 x 268 MB at N = 8192.  The GEMM-shaped substitution of gphip_solve
// (128 right-hand-side rows at a time, one launch or one dataflow hop per tile column) is latency bound for a single vector.
// Here ONE launch per triangle:
//
//   forward   x_k = W_k (b_k - sum_{c<k} L(k,c) x_c)              W_k = L_kk^-1 (explicit 128 x 128 inverses, resident)
//   backward  x_k = W_k^T (b_k - sum_{c>k} L(c,k)^T x_c)          the same schedule on mirrored block indices, tiles read transposed
//
// * a task = one 128 x 128 tile: s(I,K) = s(I,K-1) - L(I,K) x_K for K <= I-3, held in REGISTERS (32 values per lane, loaded
//   before the task's inputs exist -- L is static) by the persistent "tile" workgroups, which take tasks from a ticket in
//   column-major order (a topological order: every dependency belongs to an earlier ticket or to the chain) and prefetch the
//   next task's tile while they wait for the current one's inputs; the last link of every row sum, task (K+3, K), belongs to
//   one of TRSV_CHAIN "feeder" workgroups whose tile is resident hops ahead;
// * the chain: TRSV_CHAIN PAIRS of "chain" workgroups take the diagonal steps round-robin.  With P1_K = W_K L(K,K-1) and
//   P2_K = W_K L(K,K-2) formed once per factor (trsv_prep_kernel),
//       x_K = [ W_K s(K,K-3) - P2_K x_{K-2} ]  -  P1_K x_{K-1} :
//   the bracket needs nothing younger than two hops and is accumulated per lane BEFORE x_{K-1} arrives; what follows the
//   arrival is one half-tile product (each workgroup of a pair owns 64 of the 128 outputs), ONE 16-lane reduction and the
//   store.  The row sum the chain needs, s(K,K-3), is three hops old when it is used: the tile role is never on the critical
//   cycle (with s(K,K-2) it was: x_K -> feeder task -> chain step K+2 measured 2.0-2.7 us per hop, the arithmetic itself 0.2);
// * hand-offs carry NO flags: every 128-vector a task produces goes to its own slot of a scratch buffer that was filled with a
//   sentinel (all-ones bit pattern, a NaN no finite arithmetic produces) before the launch, written with write-through (sc1)
//   8-byte stores and polled by the consumer with sc1 loads until no word is the sentinel -- one fabric latency per hop
//   instead of payload + drain + flag + payload read.  The running sums are a chain per row block (deterministic summation
//   order, bit-identical results run to run; no atomics).
// Scratch: nrhs * 128 * Nt (Nt - 1) / 2 values for the row sums (33.5 MB per right-hand side at N = 32768) + the solution.
#pragma once
#include "gp_kernels.h"

namespace gphip {

constexpr int TRSV_MAXR = 4;        // right-hand sides per launch
constexpr int TRSV_CHAIN = 8;       // chain PAIRS (a workgroup loads 3 half tiles per step it owns: 192 KiB every 8 hops), and as many feeders
constexpr int TRSV_SPIN_LIMIT = 1 << 24;
#ifndef TRSV_TILE_BACKOFF
#define TRSV_TILE_BACKOFF 6          // s_sleep units (64 clocks: ~0.16 us) between a waiting tile task's polls of its 1-4 KiB of inputs
#endif

template <typename T>
struct TrsvArgs {
    const T* A; int R128;           // packed tile-major factor (slot 0), tile rows of the workspace (Nt + 1)
    const T* W;                     // [Nt][128 x 128] W_b = L_bb^-1, column-major, explicit zero upper triangle
    const T* P;                     // [2][Nt][128 x 128] the chain's products (trsv_prep_kernel), P[g - 1] for the block g = 1, 2 steps back:
                                    // forward P_b = W_b L(b,b-g), b >= g; backward P_b = L(b+g,b) W_b, b <= Nt - 1 - g (applied transposed)
    const T* B;                     // [nrhs][ldx] right-hand sides
    T* X;                           // [nrhs][ldx] solutions; sentinel-filled before the launch
    T* Xc;                          // [nrhs][ldx] a second copy of the solution that ONLY the next chain step polls (the tile role's
                                    // hundreds of pollers queue on X's lines at one memory channel); sentinel-filled
    T* S;                           // [I (I - 1) / 2 + K][nrhs][128] running sums s(I,K), K <= I - 3; sentinel-filled
    long ldx;
    int nt, nrhs, back;
    int dbg;                        // developer timing: 1 = the chain ignores the row sums (wrong results)
    unsigned int* ticket;           // 0xFFFFFFFF before the launch (part of the sentinel fill)
    long long* trace;               // developer timing (scripts/micro/trsv_trace.hip): 8 stamps per chain step, or null
    int* abort_flag;                // shared with the dataflow kernels (set on a spin-limit hit, never cleared here)
};

template <typename T> struct TrsvBits;
template <> struct TrsvBits<double> {
    typedef unsigned long long u;
    static __device__ __forceinline__ bool pending(double v) { return __double_as_longlong(v) == -1ll; }
};
template <> struct TrsvBits<float> {
    typedef unsigned int u;
    static __device__ __forceinline__ bool pending(float v) { return __float_as_int(v) == -1; }
};

// One wave reads nrhs x 128 values that another workgroup is about to write (or wrote long ago) into `dst` (LDS, [r][128]).
// src(r) = the global address of right-hand side r's 128 values.
// MODE 0: plain loads (the caller's right-hand side).
// MODE 1 (tile role): poll the payload with a short sleep between polls.
// MODE 2 (chain role): poll the payload back to back.
// The producer's 8-byte stores land within nanoseconds of each other in no particular order: the payload is accepted when no word
// of it is the sentinel.
template <typename T, int MODE, typename Src>
__device__ __forceinline__ void trsv_fetch(Src src, int nrhs, T* dst, int lane, int* abort_flag) {
    T va[2 * TRSV_MAXR];
    int spins = 0;
    auto give_up = [&]() -> bool {
        if ((++spins & 255) != 0) return false;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
        if (spins > TRSV_SPIN_LIMIT) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
        return false;
    };
    auto load_set = [&](T (&v)[2 * TRSV_MAXR]) {
#pragma unroll
        for (int r = 0; r < TRSV_MAXR; ++r)
            if (r < nrhs) {
                const T* p = src(r);
                if (MODE == 0) {
                    v[2 * r] = p[lane];
                    v[2 * r + 1] = p[64 + lane];
                } else {
                    v[2 * r] = __hip_atomic_load(p + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[2 * r + 1] = __hip_atomic_load(p + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
    };
    auto complete = [&](const T (&v)[2 * TRSV_MAXR]) -> bool {
        bool bad = false;
#pragma unroll
        for (int r = 0; r < TRSV_MAXR; ++r)
            if (r < nrhs) bad = bad || TrsvBits<T>::pending(v[2 * r]) || TrsvBits<T>::pending(v[2 * r + 1]);
        return __all(!bad);
    };
    load_set(va);
    if (MODE != 0) {
        // The payload itself is polled (not pipelined: a load left in flight behind the accepted one would be waited for by the next
        // s_waitcnt vmcnt(0), which is what pipelining would have saved).  MODE 1 sleeps between polls -- hundreds of workgroups
        // watch the same kilobyte -- but does NOT watch a single word first any more: a row sum is a chain of tasks, each of
        // which must see its predecessor's output within a hop (2.6 us) of the chain or the rows fall behind the chain; word +
        // payload was two fabric round trips per link (3.75 us), the payload alone is one (2.3 us with the product).
        while (__builtin_expect(!complete(va), 0)) {
            if (MODE == 1) __builtin_amdgcn_s_sleep(TRSV_TILE_BACKOFF);
            load_set(va);
            if (give_up()) break;
        }
    }
#pragma unroll
    for (int r = 0; r < TRSV_MAXR; ++r)
        if (r < nrhs) {
            dst[r * TB + lane] = va[2 * r];
            dst[r * TB + 64 + lane] = va[2 * r + 1];
        }
}

// 512 lanes share a column-major 128 x 128 tile: a lane holds the block of 8 entries along the CONTRACTED dimension --
// columns for y = M x (forward), rows for y = M^T x (backward) -- by 4 along the output dimension.  The 16 lanes of a DPP row
// (lo = tid & 15) walk the contraction, so the sum over it is a butterfly inside the row (no LDS, no bank conflicts, no
// barrier); the 32 rows of the workgroup (ob = tid >> 4) are the 32 output blocks of 4.  a[cc][rr] = M(r0 + rr, c0 + cc):
//   forward   NC = 8, NR = 4, c0 = 8 lo, r0 = 4 ob          backward   NC = 4, NR = 8, c0 = 4 ob, r0 = 8 lo
template <typename T, bool BACK> struct TrsvBlk {
    static constexpr int NC = BACK ? 4 : 8, NR = BACK ? 8 : 4;
    T a[BACK ? 4 : 8][BACK ? 8 : 4];
};
template <typename T, bool BACK>
__device__ __forceinline__ void trsv_load_tile(TrsvBlk<T, BACK>& m, const T* __restrict__ tile, int lo, int ob) {
    constexpr int NC = TrsvBlk<T, BACK>::NC, NR = TrsvBlk<T, BACK>::NR;
    const int c0 = BACK ? 4 * ob : 8 * lo, r0 = BACK ? 8 * lo : 4 * ob;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const T* p = tile + (long)(c0 + cc) * TB + r0;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) m.a[cc][rr] = p[rr];
    }
}

// Wait for every outstanding load of this wave, then tell the compiler that the tile's registers are plain values from here
// on.  Without this it protects every later use of the tile with a conservative s_waitcnt vmcnt(0) -- loads return in order,
// and inside these loops it cannot count the younger ones -- which also waits for whatever was issued AFTER the tile: the
// polls still in flight (a fabric round trip on the chain) or the next task's prefetch.
template <typename T, bool BACK>
__device__ __forceinline__ void trsv_settle(TrsvBlk<T, BACK>& m) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cc = 0; cc < TrsvBlk<T, BACK>::NC; ++cc)
#pragma unroll
        for (int rr = 0; rr < TrsvBlk<T, BACK>::NR; ++rr) asm volatile("" : "+v"(m.a[cc][rr]));
}

template <int CTRL> __device__ __forceinline__ double trsv_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float trsv_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row, the same bits in every lane: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror --
// each step pairs lanes that hold the sums of disjoint groups, and a + b = b + a
template <typename T> __device__ __forceinline__ T trsv_row16_sum(T v) {
    v += trsv_dpp<0xB1>(v);
    v += trsv_dpp<0x4E>(v);
    v += trsv_dpp<0x141>(v);
    v += trsv_dpp<0x140>(v);
    return v;
}

// y[k] = the 4 outputs of this lane's output block for ONE right-hand side (x in LDS, 128 values)
template <typename T, bool BACK>
__device__ __forceinline__ void trsv_matvec(const TrsvBlk<T, BACK>& m, const T* x, int lo, T (&y)[4]) {
    T xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = x[lo * 8 + k];
    if constexpr (!BACK) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            T acc = (T)0;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) acc = __builtin_fma(m.a[cc][rr], xv[cc], acc);
            y[rr] = acc;
        }
    } else {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            T acc = (T)0;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) acc = __builtin_fma(m.a[cc][rr], xv[rr], acc);
            y[cc] = acc;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = trsv_row16_sum(y[k]);
}

// ---- the chain's half tiles: 64 outputs x 128 contraction entries over 512 lanes = 8 along the contraction (DPP row) x 2 outputs
//   forward   a[cc][rr] = M(64 hf + 2 ob + rr, 8 lo + cc)        backward   a[cc][rr] = M(8 lo + rr, 64 hf + 2 ob + cc)
template <typename T, bool BACK> struct TrsvHalf {
    static constexpr int NC = BACK ? 2 : 8, NR = BACK ? 8 : 2;
    T a[BACK ? 2 : 8][BACK ? 8 : 2];
};
template <typename T, bool BACK>
__device__ __forceinline__ void trsv_load_half(TrsvHalf<T, BACK>& m, const T* __restrict__ tile, int lo, int ob, int hf) {
    constexpr int NC = TrsvHalf<T, BACK>::NC, NR = TrsvHalf<T, BACK>::NR;
    const int c0 = BACK ? 64 * hf + 2 * ob : 8 * lo, r0 = BACK ? 8 * lo : 64 * hf + 2 * ob;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const T* p = tile + (long)(c0 + cc) * TB + r0;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) m.a[cc][rr] = p[rr];
    }
}
template <typename T, bool BACK>
__device__ __forceinline__ void trsv_settle_half(TrsvHalf<T, BACK>& m) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cc = 0; cc < TrsvHalf<T, BACK>::NC; ++cc)
#pragma unroll
        for (int rr = 0; rr < TrsvHalf<T, BACK>::NR; ++rr) asm volatile("" : "+v"(m.a[cc][rr]));
}
// y[k] += sign * (M x) for this lane's two outputs, before the reduction over the DPP row
template <typename T, bool BACK>
__device__ __forceinline__ void trsv_half_fma(const TrsvHalf<T, BACK>& m, const T* x, int lo, bool negate, T (&y)[2]) {
    T xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = negate ? -x[lo * 8 + k] : x[lo * 8 + k];
    if constexpr (!BACK) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) y[rr] = __builtin_fma(m.a[cc][rr], xv[cc], y[rr]);
    } else {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) y[cc] = __builtin_fma(m.a[cc][rr], xv[rr], y[cc]);
    }
}

// P tiles of one direction and gap (grid = (Nt, 2): blockIdx.y = gap - 1, 256 threads): forward P_b = W_b L(b,b-gap) for b >= gap,
// backward P_b = L(b+gap,b) W_b for b <= Nt - 1 - gap; plain LDS-tiled fp64 / fp32 product, once per factor (tens of microseconds).
template <typename T>
__global__ __launch_bounds__(256) void trsv_prep_kernel(const T* __restrict__ A, int R128, const T* __restrict__ W, T* __restrict__ P, int nt, int back) {
    __shared__ T As[16][TB + 4], Bs[16][TB + 4];
    const int b = blockIdx.x, gap = (int)blockIdx.y + 1, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    if (back ? b > nt - 1 - gap : b < gap) return;
    P += (long)blockIdx.y * nt * TS;
    const T* Lt = A + tile_index(back ? b + gap : b, back ? b : b - gap, R128) * TS;
    const T* Wb = W + (long)b * TS;
    const T* Am = back ? Lt : Wb;                           // C = Am Bm
    const T* Bm = back ? Wb : Lt;
    T c[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[i][j] = (T)0;
    for (int k0 = 0; k0 < TB; k0 += 16) {
        for (int idx = tid; idx < 16 * TB; idx += 256) {
            const int kk = idx >> 7, r = idx & 127;
            As[kk][r] = Am[(long)(k0 + kk) * TB + r];                      // A(r, k0 + kk)
        }
        for (int idx = tid; idx < 16 * TB; idx += 256) {
            const int cc = idx >> 4, kk = idx & 15;
            Bs[kk][cc] = Bm[(long)cc * TB + k0 + kk];                      // B(k0 + kk, cc)
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            T av[8], bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { av[i] = As[kk][8 * ty + i]; bv[i] = Bs[kk][8 * tx + i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) c[i][j] = __builtin_fma(av[i], bv[j], c[i][j]);
        }
        __syncthreads();
    }
    T* Pb = P + (long)b * TS;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) Pb[(long)(8 * tx + j) * TB + 8 * ty + i] = c[i][j];
}

constexpr int TRSV_THREADS = 512;

template <typename T, bool BACK>
__global__ __launch_bounds__(TRSV_THREADS, 1) void trsv_dataflow_kernel(TrsvArgs<T> g) {
    constexpr int NC = TrsvBlk<T, BACK>::NC, NR = TrsvBlk<T, BACK>::NR;
    extern __shared__ double trsv_lds_raw[];
    __shared__ unsigned int s_q;
    T* xs = reinterpret_cast<T*>(trsv_lds_raw);            // [4][128] x of the source block
    T* ss = xs + TRSV_MAXR * TB;                           // [4][128] running sum coming in
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = tid & 15, ob = tid >> 4;                // block along the contraction / output block (see trsv_load_tile)
    const int nt = g.nt, nrhs = g.nrhs;
    auto blk = [&](int K) { return BACK ? nt - 1 - K : K; };                        // mirrored -> real block index
    const T* Xpoll = (g.dbg & 64) ? g.Xc : g.X;                                     // (developer timing: everybody polls the chain's copy)
    auto sslot = [&](int I, int K) { return g.S + ((long)I * (I - 1) / 2 + K) * nrhs * TB; };
    auto ltile = [&](int I, int K) {                                                // the tile that couples mirrored blocks I > K
        const int bi = blk(I), bk = blk(K);
        return g.A + tile_index(bi > bk ? bi : bk, bi > bk ? bk : bi, g.R128) * TS;
    };

    if ((int)blockIdx.x < 2 * TRSV_CHAIN) {
        // ---------------- chain role: pair c = blockIdx.x >> 1 takes the diagonal steps K = c, c + TRSV_CHAIN, ..; workgroup hf of the
        // pair computes outputs 64 hf .. 64 hf + 63 of x_K = W_K s(K,K-3) - P2_K x_{K-2} - P1_K x_{K-1}
        // (measured and dropped: a row layout -- 32 contraction entries x 1 output per lane, half the VALU operations -- is TWICE
        //  as slow: its 16 LDS reads of x per lane expose the LDS latency sixteen times; profiles/r06_trsv_chain.txt)
        TrsvHalf<T, BACK> w, p1, p2;
        const int hf = (int)blockIdx.x & 1;
        T* xs2 = ss + TRSV_MAXR * TB;                       // [4][128] x of the step before the previous one
        for (int K = (int)blockIdx.x >> 1; K < nt; K += TRSV_CHAIN) {
            const int b = blk(K);
            long long* tr = (g.trace && hf == 0) ? g.trace + (long)K * 8 : nullptr;
            long long st[8];                                // (kept in registers until the step is over: a store per stamp would put its own
            auto stamp = [&](int k) { if (tr) st[k] = wall_clock64(); };     //  completion into the next s_waitcnt vmcnt(0))
            stamp(0);
            if (tr) st[6] = clock64();
            trsv_load_half<T, BACK>(w, g.W + (long)b * TS, lo, ob, hf);
            if (K > 0) trsv_load_half<T, BACK>(p1, g.P + (long)b * TS, lo, ob, hf);
            if (K > 1) trsv_load_half<T, BACK>(p2, g.P + ((long)nt + b) * TS, lo, ob, hf);
            trsv_settle_half(w);                            // (this step's inputs are TRSV_CHAIN hops away: the wait is free)
            trsv_settle_half(p1);
            trsv_settle_half(p2);
            // early inputs: the row sum up to column K - 3 (or the right-hand side itself) and x of two steps back
            if (wave == 1) {
                if (K >= 3 && !(g.dbg & 1)) trsv_fetch<T, 2>([&](int r) { return sslot(K, K - 3) + (long)r * TB; }, nrhs, ss, lane, g.abort_flag);
                else trsv_fetch<T, 0>([&](int r) { return g.B + (long)r * g.ldx + (long)b * TB; }, nrhs, ss, lane, g.abort_flag);
            }
            // (with back-off, from the tile role's copy: only the step that is NEXT polls the chain's copy hard -- five workgroups
            //  hammering the lines of Xc stretched the hand-off from 1.25 to 1.9 us)
            if (wave == 2 && K > 1)
                trsv_fetch<T, 1>([&](int r) { return g.X + (long)r * g.ldx + (long)blk(K - 2) * TB; }, nrhs, xs2, lane, g.abort_flag);
            __syncthreads();
            stamp(2);
            T y[TRSV_MAXR][2];
#pragma unroll
            for (int r = 0; r < TRSV_MAXR; ++r) {
                y[r][0] = y[r][1] = (T)0;
                if (r < nrhs) {
                    trsv_half_fma<T, BACK>(w, ss + r * TB, lo, false, y[r]);
                    if (K > 1) trsv_half_fma<T, BACK>(p2, xs2 + r * TB, lo, true, y[r]);
                }
            }
            stamp(3);
            // the late input: x of the previous step
            if (wave == 0 && K > 0)
                trsv_fetch<T, 2>([&](int r) { return g.Xc + (long)r * g.ldx + (long)blk(K - 1) * TB; }, nrhs, xs, lane, g.abort_flag);
            stamp(1);
            __syncthreads();
            stamp(4);
#pragma unroll
            for (int r = 0; r < TRSV_MAXR; ++r)
                if (r < nrhs) {
                    if (K > 0) trsv_half_fma<T, BACK>(p1, xs + r * TB, lo, true, y[r]);
                    y[r][0] = trsv_row16_sum(y[r][0]);
                    y[r][1] = trsv_row16_sum(y[r][1]);
                    if (lo == 0) {
                        T* oc = g.Xc + (long)r * g.ldx + (long)b * TB + 64 * hf + 2 * ob;                     // the chain first
                        T* ox = g.X + (long)r * g.ldx + (long)b * TB + 64 * hf + 2 * ob;
                        __hip_atomic_store(oc, y[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(oc + 1, y[r][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(ox, y[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(ox + 1, y[r][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            stamp(5);
            if (tr) {
                st[7] = clock64();
                if (tid == 0)
                    for (int k = 0; k < 8; ++k) tr[k] = st[k];
            }
            __syncthreads();                                // (ss / xs / xs2 are rewritten by the next step)
        }
        return;
    }

    // the arithmetic of one tile task: s(I,K) = (incoming sum in ss) - L(I,K) x (in xs), stored to its slot
    auto tile_out = [&](const TrsvBlk<T, BACK>& cur, int I, int K) {
        T* out = sslot(I, K);
        for (int r = 0; r < nrhs; ++r) {
            T y[4];
            trsv_matvec<T, BACK>(cur, xs + r * TB, lo, y);
            if (lo == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(out + (long)r * TB + ob * 4 + k, ss[r * TB + ob * 4 + k] - y[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if ((int)blockIdx.x < 3 * TRSV_CHAIN) {
        // ---------------- feeder role: workgroup c takes the tasks (K + 3, K), K = c, c + TRSV_CHAIN, .. -- the last link of the row sum
        // s(K+3, K) the chain consumes three hops after x_K.  In the common ticket list that task would be picked up only after the
        // previous column's tasks have been handed out (230 workgroups, up to 250 tasks per column at N = 32768); here its tile
        // is resident hops ahead and its inputs are polled directly.
        TrsvBlk<T, BACK> cur;
        for (int K = (int)blockIdx.x - 2 * TRSV_CHAIN; K + 3 < nt; K += TRSV_CHAIN) {
            const int I = K + 3;
            trsv_load_tile<T, BACK>(cur, ltile(I, K), lo, ob);
            trsv_settle(cur);
            // (three hops of slack: polite polls, off the chain's copy)
            if (wave == 0) trsv_fetch<T, 1>([&](int r) { return g.X + (long)r * g.ldx + (long)blk(K) * TB; }, nrhs, xs, lane, g.abort_flag);
            if (wave == 1) {
                if (K > 0) trsv_fetch<T, 1>([&](int r) { return sslot(I, K - 1) + (long)r * TB; }, nrhs, ss, lane, g.abort_flag);
                else trsv_fetch<T, 0>([&](int r) { return g.B + (long)r * g.ldx + (long)blk(I) * TB; }, nrhs, ss, lane, g.abort_flag);
            }
            __syncthreads();
            tile_out(cur, I, K);
            __syncthreads();                                // (xs / ss are rewritten by the next step)
        }
        return;
    }

    // ---------------- tile role: tasks (I, K), K <= I - 4, column-major: column K holds I = K + 4 .. nt - 1
    if (nt < 5) return;
    const long ntasks = (long)(nt - 4) * (nt - 3) / 2;
    auto take = [&]() -> long {
        __syncthreads();                                    // (s_q of the previous take has been read by everybody)
        if (tid == 0) s_q = atomicAdd(g.ticket, 1u) + 1u;       // (the ticket starts at the sentinel: 0xFFFFFFFF + 1 = task 0)
        __syncthreads();
        return (long)s_q;
    };
    auto decode = [&](long q, int& I, int& K) {             // off(K) = K (nt - 4) - K (K - 1) / 2
        const double bq = (double)(2 * nt - 7);
        int k = (int)((bq - sqrt(bq * bq - 8.0 * (double)q)) * 0.5);
        if (k < 0) k = 0;
        if (k > nt - 5) k = nt - 5;
        while (k + 1 <= nt - 5 && (long)(k + 1) * (nt - 4) - (long)(k + 1) * k / 2 <= q) ++k;
        while (k > 0 && (long)k * (nt - 4) - (long)k * (k - 1) / 2 > q) --k;
        K = __builtin_amdgcn_readfirstlane(k);
        I = __builtin_amdgcn_readfirstlane(k + 4 + (int)(q - ((long)k * (nt - 4) - (long)k * (k - 1) / 2)));
    };
    TrsvBlk<T, BACK> ta, tb;
    long q = take();
    if (q >= ntasks) return;
    int I, K;
    decode(q, I, K);
    trsv_load_tile<T, BACK>(ta, ltile(I, K), lo, ob);
    auto process = [&](TrsvBlk<T, BACK>& cur, TrsvBlk<T, BACK>& nxt) -> bool {
        const long qn = take();
        int In = 0, Kn = 0;
        if (qn < ntasks) {
            decode(qn, In, Kn);
            trsv_load_tile<T, BACK>(nxt, ltile(In, Kn), lo, ob);   // in flight while this task waits for its inputs
        }
        if (wave == 0) trsv_fetch<T, 1>([&](int r) { return Xpoll + (long)r * g.ldx + (long)blk(K) * TB; }, nrhs, xs, lane, g.abort_flag);
        if (wave == 1) {
            if (K > 0) trsv_fetch<T, 1>([&](int r) { return sslot(I, K - 1) + (long)r * TB; }, nrhs, ss, lane, g.abort_flag);
            else trsv_fetch<T, 0>([&](int r) { return g.B + (long)r * g.ldx + (long)blk(I) * TB; }, nrhs, ss, lane, g.abort_flag);
        }
        trsv_settle(cur);                                   // (loads return in order: whoever saw its poll answered has both tiles)
        trsv_settle(nxt);
        __syncthreads();
        tile_out(cur, I, K);
        I = In; K = Kn;
        return qn < ntasks;                                 // (the next take()'s barrier frees xs / ss)
    };
    for (;;) {
        if (!process(ta, tb)) return;
        if (!process(tb, ta)) return;
    }
}

// A few right-hand sides for the GEMM-shaped substitution (5 .. ~100 vectors): they ride as ROWS of the 128-row scratch block V
// (column-major, leading dimension mpad).  The host used to build and move the whole zero-padded block -- 16 MB each way at
// N = 16384 for five vectors, 17 ms of a 20 ms call; now it moves the vectors themselves ([mc][npad], contiguous) and these two
// kernels scatter / gather them on the device.
template <typename T>
__global__ void rows_to_vblock_kernel(const T* __restrict__ rows, int mc, long npad, T* __restrict__ V, long mpad) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;           // point index; one thread fills column j of V
    if (j >= npad) return;
    for (long t = 0; t < mpad; ++t) V[j * mpad + t] = t < mc ? rows[t * npad + j] : (T)0;
}
template <typename T>
__global__ void vblock_to_rows_kernel(const T* __restrict__ V, long mpad, int mc, long npad, T* __restrict__ rows) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npad) return;
    for (int t = 0; t < mc; ++t) rows[(long)t * npad + j] = V[j * mpad + t];
}

// The two 64 x 64 diagonal blocks of a 128-block inverse W_b = L_bb^-1 ARE the inverses of L_bb's 64 x 64 diagonal blocks: the
// 64-block inverses the forward / backward dataflow launches substitute with (DfArgs::W, [2 Nt][64 x 64], column-major) can
// be cut out of dW after ANY fit -- also one that came from the look-ahead schedule, which only produces 128-block inverses.
template <typename T>
__global__ __launch_bounds__(256) void w128_to_w64_kernel(const T* __restrict__ W128, T* __restrict__ W64, long slot_stride = 0) {
    const int j = blockIdx.x, b = j >> 1, q = j & 1;       // 64-block j = half q of 128-block b  (blockIdx.y = batch slot)
    const T* src = W128 + (long)blockIdx.y * slot_stride + (long)b * TS + (long)(q * 64) * TB + q * 64;
    T* dst = W64 + (long)blockIdx.y * slot_stride + (long)j * 4096;
    for (int idx = threadIdx.x; idx < 4096; idx += 256) {
        const int c = idx >> 6, r = idx & 63;
        dst[idx] = src[(long)c * TB + r];
    }
}

constexpr size_t trsv_lds_bytes(size_t es) { return (size_t)(3 * TRSV_MAXR * TB) * es; }

}  // namespace gphip
