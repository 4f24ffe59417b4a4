// Shader clock seen by a lone workgroup on an otherwise idle GPU vs. under load (DVFS check).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* out, int iters) {
    double a = threadIdx.x * 1e-9, b = 1.0000001;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) a = __builtin_fma(a, b, 1e-12);
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)(a * 0); }
}
__global__ void burn(double* x, int iters) {
    double a = x[threadIdx.x];
    for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, 1e-12);
    x[threadIdx.x] = a;
}
int main() {
    long long* d; hipMalloc(&d, 64); double* x; hipMalloc(&x, 8 * 256); hipMemset(x, 0, 8 * 256);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate attr: %d kHz\n", wc);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 20000);
        hipDeviceSynchronize();
        long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("idle:  %lld shader cycles in %lld wall ticks -> %.0f MHz (if wall = %d kHz); %.1f cycles/fma\n", h[0], h[1],
               (double)h[0] / h[1] * wc / 1e3, wc, (double)h[0] / 20000);
    }
    hipStream_t s2; hipStreamCreate(&s2);
    hipLaunchKernelGGL(burn, dim3(4096), dim3(256), 0, s2, x, 3000000);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 20000);
        hipStreamSynchronize(0);
        long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("busy:  %lld shader cycles in %lld wall ticks -> %.0f MHz\n", h[0], h[1], (double)h[0] / h[1] * wc / 1e3);
    }
    hipDeviceSynchronize();
    return 0;
}
