// fp64 VALU dependent-chain latency vs ILP on gfx950 (one wave on an idle CU), plus rsqrt / sqrt / rcp cost.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void fma_chain(long long* out, double* sink, int iters) {
    double a[ILP];
    for (int k = 0; k < ILP; ++k) a[k] = threadIdx.x * 1e-9 + k;
    const double b = 1.0000001;
    long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) a[k] = __builtin_fma(a[k], b, 1e-12);
        }
    }
    long long c1 = clock64();
    double s = 0; for (int k = 0; k < ILP; ++k) s += a[k];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[0] = c1 - c0;
}
template <int OP>
__global__ void op_chain(long long* out, double* sink, int iters) {
    double a = 1.5 + threadIdx.x * 1e-3;
    long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) a = rsqrt(a) + 1.0;
            if (OP == 1) a = sqrt(a) + 1.0;
            if (OP == 2) a = 1.0 / a + 1.0;
            if (OP == 3) a = __builtin_amdgcn_rsq(a) + 1.0;     // raw v_rsq_f64
            if (OP == 4) a = __builtin_amdgcn_rcp(a) + 1.0;     // raw v_rcp_f64
        }
    }
    long long c1 = clock64();
    sink[threadIdx.x] = a;
    if (threadIdx.x == 0) out[0] = c1 - c0;
}
__global__ void f32_chain(long long* out, float* sink, int iters) {
    float a = threadIdx.x * 1e-6f;
    long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = __builtin_fmaf(a, 1.0000001f, 1e-7f);
    }
    long long c1 = clock64();
    sink[threadIdx.x] = a;
    if (threadIdx.x == 0) out[0] = c1 - c0;
}
int main() {
    long long* d; hipMalloc(&d, 64); double* s; hipMalloc(&s, 8 * 64);
    long long h; const int it = 20000;
#define RUN(K, label, per) hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, d, s, it); hipDeviceSynchronize(); \
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf("%-28s %.2f cycles per %s\n", label, (double)h / it / 16, per);
    RUN(fma_chain<1>, "fp64 fma ILP=1", "iteration (1 fma)");
    RUN(fma_chain<2>, "fp64 fma ILP=2", "iteration (2 fma)");
    RUN(fma_chain<4>, "fp64 fma ILP=4", "iteration (4 fma)");
    RUN(fma_chain<8>, "fp64 fma ILP=8", "iteration (8 fma)");
    RUN(fma_chain<16>, "fp64 fma ILP=16", "iteration (16 fma)");
    RUN(op_chain<0>, "rsqrt(double)+add (x2)", "iteration");
    RUN(op_chain<1>, "sqrt(double)+add", "iteration");
    RUN(op_chain<2>, "1/x (double)+add", "iteration");
    RUN(op_chain<3>, "v_rsq_f64 raw +add", "iteration");
    RUN(op_chain<4>, "v_rcp_f64 raw +add", "iteration");
    hipLaunchKernelGGL(f32_chain, dim3(1), dim3(64), 0, 0, d, (float*)s, it); hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf("%-28s %.2f cycles per fma\n", "fp32 fma ILP=1", (double)h / it / 16);
    return 0;
}
