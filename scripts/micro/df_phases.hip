// Per-task timeline of chol_dataflow_kernel<double, 64, 2, 2> on a synthetic SPD matrix in the packed tile-major layout,
// WITH the phase stamps of the potrf body inside the diagonal tasks (developer tool; wall clock = 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/df_phases scripts/micro/df_phases.hip && /tmp/df_phases 64
// arguments: number of 128-tile columns (64 -> N = 8192: the loaded regime; 8 -> N = 1024: the chain alone) [LDS KiB per workgroup]
#define GPHIP_TIMING 1
#include "../../bayesianinference_amd/csrc/gp_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gphip;
int main(int argc, char** argv) {
    const int Nt = argc > 1 ? atoi(argv[1]) : 64, n = Nt * TB, nd = n / 64, R = nd + 1, R128 = Nt + 1;
    const long slot_elems = (long)R128 * (R128 + 1) / 2 * TS;
    std::vector<double> A((size_t)slot_elems, 0.0);
    auto at = [&](int i, int j) -> double& {               // element (i, j), i >= j, of the bordered matrix
        return A[(size_t)(tile_index(i / TB, j / TB, R128) * TS + (long)(j % TB) * TB + (i % TB))];
    };
    for (int j = 0; j < n; ++j) {
        const int hi = std::min(n, j + 400);               // (banded fill is enough: exp(-d^2/1800) < 1e-38 beyond 400)
        for (int i = j; i < hi; ++i) at(i, j) = std::exp(-0.5 * (double)(i - j) * (i - j) / 900.0) + (i == j ? 0.1 : 0.0);
    }
    for (int j = 0; j < n; ++j) at(n, j) = std::sin(0.01 * j);                 // the rhs row
    for (int r = 1; r < TB; ++r) at(n + r, n + r) = 1.0;
    const long ntask = (long)R * (R + 1) / 2;
    double *dA, *dW, *dP, *dS, *dD; int *dI, *dF; unsigned long long* dT; long long *dTr, *dSt;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, (size_t)Nt * TB * TB * 8); hipMalloc(&dP, nd * 8); hipMalloc(&dS, 128); hipMalloc(&dD, (size_t)nd * 1024 * 8);
    hipMalloc(&dI, 4); hipMalloc(&dF, (size_t)R * R * 4); hipMalloc(&dT, 16); hipMalloc(&dTr, ntask * 64); hipMalloc(&dSt, ntask * 512);
    double sp[SLOTP] = {1.0, 0.1, 0.0, 1e-14};
    hipMemcpy(dS, sp, sizeof sp, hipMemcpyHostToDevice); hipMemset(dI, 0, 4); hipMemset(dF, 0, (size_t)R * R * 4); hipMemset(dT, 0, 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dSt, sizeof dSt);
    size_t lds = std::max(df_lds_bytes<double, 64, 2>(), DF_XXF_LDS);
    if (argc > 2 && (size_t)atoi(argv[2]) * 1024 > lds) lds = (size_t)atoi(argv[2]) * 1024;      // > 80 KiB: one workgroup per CU
    auto kern = chol_dataflow_kernel<double, 64, 2, 2, false>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    std::vector<long long> tr((size_t)ntask * 8), st((size_t)ntask * 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemset(dTr, 0, ntask * 64); hipMemset(dSt, 0, ntask * 512);
        DfArgs<double> g{};
        g.A = dA; g.bstride = slot_elems; g.R128 = R128; g.c0 = 0; g.W = dW; g.w_bstride = (long)Nt * TB * TB; g.partial = dP;
        g.p_bstride = nd; g.info = dI; g.slotp = dS; g.flags = dF; g.f_bstride = (long)R * R; g.ticket = dT;
        g.ticket_base = (unsigned long long)rep * ntask; g.abort_flag = (int*)(dT + 1); g.nd = nd; g.nslots = 1; g.epoch = rep + 1;
        g.trace = dTr; g.D = dD; g.d_bstride = (long)nd * 1024;
        hipLaunchKernelGGL(kern, dim3((unsigned)ntask), dim3(256), lds, 0, g, ThetaPack{});
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    }
    hipMemcpy(tr.data(), dTr, ntask * 64, hipMemcpyDeviceToHost);
    hipMemcpy(st.data(), dSt, ntask * 512, hipMemcpyDeviceToHost);
    std::vector<long> block_of((size_t)ntask, -1);
    for (long b = 0; b < ntask; ++b) { const long t = st[(size_t)b * 64 + 63]; if (t >= 0 && t < ntask) block_of[(size_t)t] = b; }
    auto off = [&](int j) { return (long)j * R - (long)j * (j - 1) / 2; };
    const long long t0 = tr[0];
    auto us = [&](long long v) { return v ? (v - t0) / 100.0 : -1.0; };
    auto hop = [&](int j) {
        const long long* d = &tr[(size_t)off(j) * 8]; const long long* p = &tr[(size_t)off(j - 1) * 8];
        const long bp = block_of[(size_t)off(j - 1)];
        // ready(j-1,j-1) goes out INSIDE the potrf body (stamp 35 of that workgroup), before its L tile
        // (the chain's own hand-over is earlier still: stamp 32, after the L tile + diagonal inverses and their flag)
        const double t = (bp >= 0 && st[(size_t)bp * 64 + 32]) ? us(st[(size_t)bp * 64 + 32]) : us(p[4]);
        printf("hop diag(%d): prev pub %.1f | flags seen +%.1f | solve+store+publish +%.1f | slab +%.1f | packed +%.1f | potrf call end +%.1f\n",
               j, t, us(d[5]) - t, us(d[7]) - t, us(d[6]) - t, us(d[2]) - t, us(d[3]) - t);
        const long b = block_of[(size_t)off(j)];
        if (b < 0) return;
        const long long* s = &st[(size_t)b * 64];
        auto du = [&](int a, int c) { return (s[c] - s[a]) / 100.0; };
        printf("   potrf64 phases (us): entry %.2f |", du(1, 2));
        for (int q = 0; q < 4; ++q) printf(" panel %d: upd+elim %.2f%s", q, du(2 + 3 * q, 3 + 3 * q), q < 3 ? "," : " |");
        // (round 5: the diagonal inverses come out of the elimination; order = logdet, block inverses, W store, publish, L store)
        printf(" L + diagonal inverses + chain flag %.2f | entry..chain flag %.2f || off the chain: block inv %.2f | W store %.2f | publish %.2f   (call overhead: %.2f before, %.2f after)\n",
               du(12, 32), du(1, 32), du(32, 33), du(33, 34), du(34, 35), (s[1] - d[2]) / 100.0, (d[3] - s[36]) / 100.0);
    };
    for (int j : {nd / 8, nd / 4, nd / 2, nd / 2 + 1, nd / 2 + 2, nd / 2 + 3, 3 * nd / 4, nd - 3}) if (j >= 1 && j < nd) hop(j);
    long long tend = 0; for (long q = 0; q < ntask; ++q) for (int k = 0; k < 8; ++k) tend = std::max(tend, tr[(size_t)q * 8 + k]);
    printf("kernel span %.2f us = %.2f us per 64 columns\n", us(tend), us(tend) / nd);
    int ab; hipMemcpy(&ab, dT + 1, 4, hipMemcpyDeviceToHost); int info; hipMemcpy(&info, dI, 4, hipMemcpyDeviceToHost);
    printf("abort %d info %d\n", ab, info);
    return 0;
}
