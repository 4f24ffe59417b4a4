// Timing experiments on the trailing-SYRK kernel (developer tool): where do the idle MFMA cycles go?
// Build with -DGP_DIAG=<bits> (see gp_kernels.h): 0 real kernel, 1 no DMA after the first two stages,
// 2 no C load/store, 4 no main-loop barrier, and combinations.  Results are garbage for GP_DIAG != 0.
#include "../../bayesianinference_amd/csrc/gp_kernels.h"
#include <cstdio>
#include <cstdlib>
using namespace gphip;
__global__ void fill_random(double* a, long n) {      // realistic bit toggling: zeros run at higher clocks
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long z = (unsigned long)i * 0x9E3779B97F4A7C15ul + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ul; z = (z ^ (z >> 27)) * 0x94D049BB133111EBul; z ^= z >> 31;
        a[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-2;
    }
}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 192, P = 4;      // trailing tile rows, panel width in tiles
    const int R = H + P;
    const long ld = (long)R * TB;
    double* A;
    if (hipMalloc(&A, ld * ld * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    if (argc > 2 && atoi(argv[2]) == 0) hipMemset(A, 0, ld * ld * 8);
    else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, A, ld * ld);
    GemmArgs<double> g{};
    g.C = A; g.ldc = ld; g.A = A; g.lda = ld; g.B = A; g.ldb = ld;
    g.K = P * TB; g.r0 = P; g.r1 = R; g.c0 = P; g.c1 = R; g.tri = 1; g.nrect = 0;
    g.ntiles = H * (H + 1) / 2; g.swizzle = 1;
    auto kern = gemm_nt_kernel<double, 0, 2, 2, 2>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (2 * STAGE_BYTES));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(g.ntiles), dim3(256), (2 * STAGE_BYTES), 0, g);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * TB * TB * (double)g.K * g.ntiles;
        if (rep) printf("GP_DIAG=%d H=%d tiles=%d rounds=%.2f  %.3f ms  %.2f TFLOP/s\n", GP_DIAG, H, g.ntiles, g.ntiles / 512.0, ms, fl / ms * 1e-9);
    }
    return 0;
}
