// Streaming-store rate of "one contiguous chunk per workgroup" kernels by chunk size and workgroup size (round 5: the kernel
// build writes one 128 KiB tile per 256-thread workgroup and reaches 0.72 of 8 TB/s; torch's fill reaches 0.86 -- which
// geometry does the difference come from?).   hipcc --offload-arch=gfx950 -O3 -o store_sweep scripts/micro/store_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
// workgroup b writes bytes [b * chunk, (b + 1) * chunk): per iteration the workgroup's T threads write T * 16 contiguous bytes
template <int T>
__global__ __launch_bounds__(T) void chunk_store(dv2* out, int iters) {
    dv2* o = out + (long)blockIdx.x * iters * T + threadIdx.x;
    const dv2 v = {1.0 + threadIdx.x, 2.0};
    for (int k = 0; k < iters; ++k) o[(long)k * T] = v;
}
// the same bytes per workgroup, but each WAVE owns a contiguous quarter (the kernel build's original order)
template <int T>
__global__ __launch_bounds__(T) void wave_chunk_store(dv2* out, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = T / 64;
    dv2* o = out + (long)blockIdx.x * iters * T + (long)wave * iters * 64 + lane;
    const dv2 v = {1.0 + threadIdx.x, 2.0};
    for (int k = 0; k < iters; ++k) o[(long)k * 64] = v;
    (void)nw;
}
int main() {
    const size_t bytes = (size_t)32896 * 131072;          // 4.31 GB: the packed lower triangle of N = 32768
    dv2* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipMemset(d, 0, bytes);
    for (int rep = 0; rep < 2; ++rep)
        for (int kib : {8, 16, 32, 64, 128, 256, 512}) {
#define RUN(KERNEL, T, name)                                                                                         \
            {                                                                                                        \
                const int iters = kib * 1024 / (T * 16);                                                             \
                if (iters >= 1) {                                                                                    \
                    const unsigned grid = (unsigned)(bytes / ((size_t)kib * 1024));                                  \
                    float best = 1e9;                                                                                \
                    for (int r = 0; r < 3; ++r) {                                                                    \
                        hipEventRecord(e0);                                                                          \
                        hipLaunchKernelGGL((KERNEL<T>), dim3(grid), dim3(T), 0, 0, d, iters);                        \
                        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);               \
                        best = ms < best ? ms : best;                                                                \
                    }                                                                                                \
                    printf("%s T=%d chunk=%d KiB: %.3f ms = %.2f TB/s = %.3f of 8\n", name, T, kib, best, bytes / best / 1e9, bytes / best / 8e9); \
                }                                                                                                    \
            }
            RUN(chunk_store, 256, "chunk")
            RUN(chunk_store, 512, "chunk")
            RUN(chunk_store, 1024, "chunk")
            RUN(wave_chunk_store, 256, "wave-chunk")
        }
    hipEventRecord(e0); hipMemsetAsync(d, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemsetAsync: %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    hipFree(d);
    return 0;
}
