// Per-task timeline of chol_dataflow_kernel<double, DF_TBX, (DF_TBX == 128 ? 1 : 2), (DF_TBX == 128 ? 4 : 2)> on a synthetic SPD matrix (developer tool).
// Prints, for the tasks on the critical path, wall-clock stamps (100 MHz s_memrealtime) in us.
#include "../../bayesianinference_amd/csrc/gp_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
using namespace gphip;
#ifndef DF_TBX
#define DF_TBX 128
#endif
int main(int argc, char** argv) {
    const int Nt = argc > 1 ? atoi(argv[1]) : 8, n = Nt * TB, nd = n / DF_TBX, R = nd + 1;   // Nt counts 128-tiles
    const long ld = (long)(Nt + 1) * TB;
    std::vector<double> A((size_t)ld * ld, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) A[(size_t)j * ld + i] = std::exp(-0.5 * (double)(i - j) * (i - j) / 900.0) + (i == j ? 0.1 : 0.0);
    for (int j = 0; j < n; ++j) A[(size_t)j * ld + n] = std::sin(0.01 * j);     // one rhs row
    for (int r = 1; r < TB; ++r) A[(size_t)(n + r) * ld + n + r] = 1.0;
    const long ntask = (long)R * (R + 1) / 2;
    double *dA, *dW, *dP, *dS; int *dI, *dF; unsigned long long* dT; long long* dTr;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, (size_t)Nt * TB * TB * 8); hipMalloc(&dP, nd * 8); hipMalloc(&dS, 64);
    hipMalloc(&dI, 4); hipMalloc(&dF, R * R * 4); hipMalloc(&dT, 16); hipMalloc(&dTr, ntask * 64);
    double sp[8] = {1.0, 0.1, 0.0, 1e-14, 0, 0, 0, 0};
    hipMemcpy(dS, sp, 64, hipMemcpyHostToDevice); hipMemset(dI, 0, 4); hipMemset(dF, 0, R * R * 4); hipMemset(dT, 0, 16);
    const size_t lds = df_lds_bytes<double, DF_TBX, (DF_TBX == 128 ? 4 : 2)>();
    hipFuncSetAttribute(reinterpret_cast<const void*>(chol_dataflow_kernel<double, DF_TBX, (DF_TBX == 128 ? 1 : 2), (DF_TBX == 128 ? 4 : 2)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<long long> tr((size_t)ntask * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemset(dTr, 0, ntask * 64);
        DfArgs<double> g{};
        g.A = dA; g.ld = ld; g.bstride = ld * ld; g.W = dW; g.w_bstride = (long)Nt * TB * TB; g.partial = dP; g.info = dI;
        g.slotp = dS; g.flags = dF; g.f_bstride = (long)R * R; g.ticket = dT; g.ticket_base = (unsigned long long)rep * ntask;
        g.abort_flag = (int*)(dT + 1); g.nd = nd; g.nslots = 1; g.epoch = rep + 1; g.trace = dTr;
        hipLaunchKernelGGL((chol_dataflow_kernel<double, DF_TBX, (DF_TBX == 128 ? 1 : 2), (DF_TBX == 128 ? 4 : 2)>), dim3((unsigned)ntask), dim3(256), lds, 0, g, ThetaPack{});
        hipDeviceSynchronize();
    }
    hipMemcpy(tr.data(), dTr, ntask * 64, hipMemcpyDeviceToHost);
    auto off = [&](int j) { return (long)j * R - (long)j * (j - 1) / 2; };
    const long long t0 = tr[0];
    auto us = [&](long long v) { return v ? (v - t0) / 100.0 : -1.0; };
    printf("task(i,j): start | deps+acc done | [trsm: diag flag seen] | core begin | core end | published   (us)\n");
    for (int j = 0; j < std::min(nd, 6); ++j) {
        const long long* d = &tr[(size_t)off(j) * 8];
        printf("diag(%d,%d): %.2f | last slab flag %.2f | %.2f | - | %.2f | %.2f | %.2f\n", j, j, us(d[0]), us(d[6]), us(d[1]), us(d[2]), us(d[3]), us(d[4]));
        const long long* t = &tr[(size_t)(off(j) + 1) * 8];
        printf("trsm(%d,%d): %.2f | last slab flag %.2f | %.2f | %.2f | %.2f | %.2f | %.2f\n", j + 1, j, us(t[0]), us(t[6]), us(t[1]), us(t[5]), us(t[2]), us(t[3]), us(t[4]));
    }
    if (nd > 40) for (int j = 60; j < 64; ++j) {      // the hop of a mid diagonal task, split: (times relative to the previous diagonal's publish)
        const long long* d = &tr[(size_t)off(j) * 8]; const long long* p = &tr[(size_t)off(j - 1) * 8];
        const double t = us(p[4]);
        printf("hop diag(%d): prev pub %.1f | flags seen +%.1f | solve+store+publish +%.1f | slab +%.1f | packed +%.1f | potrf +%.1f | pub +%.1f   (pre flag owner published at %+.1f)\n",
               j, t, us(d[5]) - t, us(d[7]) - t, us(d[6]) - t, us(d[2]) - t, us(d[3]) - t, us(d[4]) - t, us(tr[(size_t)(off(j - 1) + 1) * 8 + 4]) - t);
    }
    if (nd > 40) for (int j = 30; j < 33; ++j) {
        const long long* d = &tr[(size_t)off(j) * 8]; const long long* x = &tr[(size_t)(off(j) + 1) * 8];
        printf("mid diag(%d): start %.1f lastflag %.1f accdone %.1f corebeg %.1f coreend %.1f pub %.1f | trsm(%d,%d): start %.1f accdone %.1f flag %.1f end %.1f pub %.1f\n", j, us(d[0]), us(d[6]), us(d[1]), us(d[2]), us(d[3]), us(d[4]), j + 1, j, us(x[0]), us(x[1]), us(x[5]), us(x[3]), us(x[4]));
    }
    // pace of the diagonal: potrf end of every 8th diagonal task
    for (int j = 0; j < nd; j += 8) { const long long* d = &tr[(size_t)off(j) * 8]; printf("diag %d: start %.1f flagseen %.1f accdone %.1f potrf_end %.1f\n", j, us(d[0]), us(d[6]), us(d[1]), us(d[3])); }
    long long tend = 0; for (long q = 0; q < ntask; ++q) for (int k = 0; k < 8; ++k) tend = std::max(tend, tr[(size_t)q * 8 + k]);
    printf("kernel span %.2f us\n", us(tend));
    int ab; hipMemcpy(&ab, dT + 1, 4, hipMemcpyDeviceToHost); printf("abort %d\n", ab);
    return 0;
}
