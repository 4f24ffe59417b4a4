// Issue rates of the fp64 VALU and DPP instructions the single-vector substitution uses (developer tool): shader clocks per
// wave64 instruction with 1 or 2 waves per SIMD, independent and dependent chains.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/valu_rate scripts/micro/valu_rate.hip && scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, long long* clk, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = seed * 0.5, c = seed * 0.25;
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (MODE == 0) {            // 8 independent FMA chains
            a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
            a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
        } else if (MODE == 1) {     // one dependent FMA chain
            a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c);
            a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c); a0 = __builtin_fma(a0, b, c);
        } else if (MODE == 2) {     // 8 independent adds
            a0 += b; a1 += b; a2 += b; a3 += b; a4 += b; a5 += b; a6 += b; a7 += b;
        } else if (MODE == 3) {     // 4 x (dpp pair + add): the butterfly step
#define STEP(v) { const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true); \
                  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true); v += __hiloint2double(hi, lo); }
            STEP(a0) STEP(a1) STEP(a2) STEP(a3)
        } else if (MODE == 4) {     // 4 independent chains of 2 dependent FMAs (distance 4)
            a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
            a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 512 * 8 * 4); hipMalloc(&clk, 64);
    const char* names[] = {"8 independent v_fma_f64", "8 dependent v_fma_f64", "8 independent v_add_f64", "4 x (2 v_mov_dpp + v_add_f64)", "4 chains x 2 dependent v_fma_f64"};
    for (int threads : {256, 512}) {
        for (int m = 0; m < 5; ++m) {
            long long c = 0;
            for (int rep = 0; rep < 3; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001);
                if (m == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, out, clk, 1.0000001);
                hipDeviceSynchronize();
                hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            }
            const int per = m == 3 ? 12 : 8;
            printf("%d threads (%d wave%s per SIMD): %-34s %6.2f clocks per iteration = %5.2f per instruction\n", threads, threads / 256,
                   threads > 256 ? "s" : "", names[m], c / 256.0, c / 256.0 / per);
        }
    }
    return 0;
}
