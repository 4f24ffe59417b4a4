// Developer A/B (one process, interleaved rounds): the trailing-SYRK kernel on (1) the round-2 kernel + column-major
// workspace, (2) the current kernel on a column-major workspace, (3) the current kernel on the packed tile-major workspace.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<dir with the round-2 gp_kernels.h as old_gp_kernels.h> -o layout_ab layout_ab.hip
#define gphip gphip_old
#include "old_gp_kernels.h"
#undef gphip
#undef GP_STAMP
#undef GP_DIAG
#undef GP_DF_PRIO
#include "../../bayesianinference_amd/csrc/gp_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void fill_random(double* a, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long z = (unsigned long)i * 0x9E3779B97F4A7C15ul + 12345;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ul; z = (z ^ (z >> 27)) * 0x94D049BB133111EBul; z ^= z >> 31;
        a[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-2;
    }
}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 192, P = argc > 2 ? atoi(argv[2]) : 4;
    const int R = H + P;
    const long ld = (long)R * 128 + 128;
    const long tiled_elems = (long)R * (R + 1) / 2 * gphip::TS;
    double *A, *Tl;
    if (hipMalloc(&A, ld * ld * 8) != hipSuccess || hipMalloc(&Tl, tiled_elems * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, A, ld * ld);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, Tl, tiled_elems);
    const int ntiles = H * (H + 1) / 2;
    gphip_old::GemmArgs<double> go{};
    go.C = A; go.ldc = ld; go.A = A; go.lda = ld; go.B = A; go.ldb = ld;
    go.K = P * 128; go.r0 = P; go.r1 = R; go.c0 = P; go.c1 = R; go.tri = 1; go.nrect = 0; go.ntiles = ntiles; go.swizzle = 1;
    go.thin_row = -1;
    gphip::GemmArgs<double> gs{}, gt{};
    gs.C = A; gs.ldc = ld; gs.A = A; gs.lda = ld; gs.B = A; gs.ldb = ld;
    gs.K = P * 128; gs.r0 = P; gs.r1 = R; gs.c0 = P; gs.c1 = R; gs.tri = 1; gs.nrect = 0; gs.ntiles = ntiles; gs.swizzle = 1;
    gs.thin_row = -1;
    gt = gs;
    gt.C = Tl; gt.A = Tl; gt.B = Tl; gt.c_R = gt.a_R = gt.b_R = R; gt.a_k0 = gt.b_k0 = 0;
    auto ko = gphip_old::gemm_nt_kernel<double, 0, 2, 2, 2>;
    auto kn = gphip::gemm_nt_kernel<double, 0, 2, 2, 2>;
    const int lds = 2 * gphip::STAGE_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void*>(ko), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    gphip::GemmArgs<double> gc = gs, gab = gs;           // mixed: only C tiled / only the operands tiled
    gc.C = Tl; gc.c_R = R;
    gab.A = Tl; gab.B = Tl; gab.a_R = gab.b_R = R;
    auto k7 = gphip::gemm_nt_kernel<double, 0, 2, 2, 2>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k7), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<float> t[6];
    for (int rep = 0; rep < 8; ++rep)
        for (int v = 0; v < 6; ++v) {
            hipEventRecord(e0, 0);
            if (v == 0) hipLaunchKernelGGL(ko, dim3(ntiles), dim3(256), lds, 0, go);
            else if (v == 1) hipLaunchKernelGGL(kn, dim3(ntiles), dim3(256), lds, 0, gs);
            else if (v == 2) hipLaunchKernelGGL(kn, dim3(ntiles), dim3(256), lds, 0, gt);
            else if (v == 3) hipLaunchKernelGGL(kn, dim3(ntiles), dim3(256), lds, 0, gc);
            else if (v == 4) hipLaunchKernelGGL(kn, dim3(ntiles), dim3(256), lds, 0, gab);
            else hipLaunchKernelGGL(k7, dim3(ntiles), dim3(256), lds, 0, gt);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) t[v].push_back(ms);
        }
    const double fl = 2.0 * 128 * 128 * (double)(P * 128) * ntiles;
    const char* names[6] = {"r02 kernel, column-major", "new kernel, column-major", "new kernel, tile-major", "new, only C tile-major",
                            "new, only A/B tile-major", "new kernel, tile-major (again)"};
    for (int v = 0; v < 6; ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("H=%d P=%d pad=%d  %-26s median %.3f ms  %.2f TFLOP/s (min %.3f max %.3f)\n", H, P, (int)GP_TILE_PAD, names[v], med,
               fl / med * 1e-9, t[v].front(), t[v].back());
    }
    return 0;
}
