// Per-step timeline of the chain of trsv_dataflow_kernel<double> (csrc/gp_trsv.h) on a synthetic factor in the packed
// tile-major layout (developer tool; wall clock = 100 MHz s_memrealtime).  Values are irrelevant to the timing: L = small
// random entries with a unit diagonal, W = identity.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/trsv_trace scripts/micro/trsv_trace.hip && /tmp/trsv_trace 64 1
// arguments: 128-tile columns (64 -> N = 8192) [right-hand sides] [dbg bits] [workgroups (default: CU count)]
#include "../../bayesianinference_amd/csrc/gp_trsv.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gphip;
int main(int argc, char** argv) {
    const int Nt = argc > 1 ? atoi(argv[1]) : 64, nrhs = argc > 2 ? atoi(argv[2]) : 1, dbg = argc > 3 ? atoi(argv[3]) : 0;
    const int R128 = Nt + 1;
    const long npad = (long)Nt * TB, slot_elems = (long)R128 * (R128 + 1) / 2 * TS;
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int nwg = argc > 4 ? atoi(argv[4]) : ncu;
    std::vector<double> A((size_t)slot_elems), W((size_t)Nt * TS, 0.0), B((size_t)nrhs * npad, 1.0);
    unsigned s = 12345u;
    for (auto& v : A) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 * 1e-4; }
    for (int b = 0; b < Nt; ++b)
        for (int i = 0; i < TB; ++i) W[(size_t)b * TS + (size_t)i * TB + i] = 1.0;
    double *dA, *dW, *dB, *dX, *dXc, *dS, *dP; unsigned int* dT; int* dAb; long long* dTr;
    const size_t sbytes = (size_t)nrhs * TB * ((size_t)Nt * (Nt - 1) / 2 + 1) * 8;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, W.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dX, B.size() * 8);
    hipMalloc(&dP, 2 * W.size() * 8); hipMemcpy(dP, A.data(), 2 * W.size() * 8, hipMemcpyHostToDevice);      // (any small numbers)
    hipMalloc(&dXc, B.size() * 8); hipMalloc(&dS, sbytes); hipMalloc(&dT, 256); hipMalloc(&dAb, 256); hipMalloc(&dTr, (size_t)Nt * 64);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice); hipMemset(dAb, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<long long> tr((size_t)Nt * 8);
    for (int back = 0; back < 2; ++back) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(dX, 0xFF, B.size() * 8); hipMemset(dXc, 0xFF, B.size() * 8); hipMemset(dS, 0xFF, sbytes); hipMemset(dT, 0xFF, 256);
            hipMemset(dTr, 0, (size_t)Nt * 64);
            TrsvArgs<double> g{};
            g.A = dA; g.R128 = R128; g.W = dW; g.P = dP; g.B = dB; g.X = dX; g.Xc = dXc; g.S = dS; g.ldx = npad; g.nt = Nt; g.nrhs = nrhs; g.back = back;
            g.dbg = dbg; g.ticket = dT; g.abort_flag = dAb; g.trace = dTr;
            const long ntasks = Nt >= 5 ? (long)(Nt - 4) * (Nt - 3) / 2 : 0;
            const long grid = std::min<long>(nwg, 3 * TRSV_CHAIN + ntasks);
            hipEventRecord(e0, 0);
            if (back) hipLaunchKernelGGL((trsv_dataflow_kernel<double, true>), dim3((unsigned)grid), dim3(TRSV_THREADS), trsv_lds_bytes(8), 0, g);
            else hipLaunchKernelGGL((trsv_dataflow_kernel<double, false>), dim3((unsigned)grid), dim3(TRSV_THREADS), trsv_lds_bytes(8), 0, g);
            hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        hipMemcpy(tr.data(), dTr, (size_t)Nt * 64, hipMemcpyDeviceToHost);
        int ab = 0; hipMemcpy(&ab, dAb, 4, hipMemcpyDeviceToHost);
        printf("%s: Nt=%d nrhs=%d dbg=%d wgs=%d: %.3f ms = %.2f us per step (abort %d)\n", back ? "backward" : "forward", Nt, nrhs, dbg, nwg, best,
               best * 1e3 / Nt, ab);
        auto us = [&](long long a, long long b) { return (a - b) / 100.0; };
        double acc[8] = {0};
        int cnt = 0;
        for (int K = 3; K < Nt; ++K) {
            const long long* t = &tr[(size_t)K * 8]; const long long* p = &tr[(size_t)(K - 1) * 8];
            // stamps: 0 step start, 2 early inputs in LDS, 3 early sums done, 1 x of the previous step fetched, 4 barrier passed, 5 stored
            const double v[6] = {us(t[2], p[5]), us(t[3], t[2]), us(t[1], p[5]), us(t[4], t[1]), us(t[5], t[4]), us(t[5], p[5])};
            if (K % (Nt / 8 > 0 ? Nt / 8 : 1) == 3)
                printf("  step %3d: early inputs in %+6.2f (rel. prev store) | early sums %5.2f | x seen %+6.2f (rel. prev store) | barrier %5.2f | P1 x + reduction + store %5.2f | step start was %7.2f before x seen\n",
                       K, v[0], v[1], v[2], v[3], v[4], us(t[1], t[0]));
            for (int i = 0; i < 6; ++i) acc[i] += v[i];
            acc[6] += (double)(t[7] - t[6]) / (double)(t[5] - t[0]) * 100.0;          // shader clocks per 10 ns tick -> MHz
            ++cnt;
        }
        printf("  mean over %d steps: early inputs in %+.2f (rel. prev store) | early sums %.2f | hand-off (prev stored -> x seen) %.2f | barrier %.2f | "
               "P1 x + reduction + store %.2f || store-to-store %.2f us | shader clock %.0f MHz\n", cnt, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt,
               acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt);
    }
    return 0;
}
