// Does a CU-masked stream work on this stack?  A kernel of 4096 one-per-CU workgroups (100 KB LDS each) that each spin ~100 us:
// 256 CUs -> 16 rounds, 240 CUs -> 18 rounds.  Then: latency of a 1-workgroup kernel on a second stream while the big one runs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) {
    extern __shared__ double lds[];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (ticks < 0) lds[threadIdx.x] = 0;
}
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipStream_t full, masked, small;
    hipStreamCreateWithFlags(&full, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&small, hipStreamNonBlocking);
    std::vector<uint32_t> mask(8, 0xFFFFFFFFu);
    mask[0] = 0xFFFF0000u;                       // 16 CUs off
    hipError_t e = hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data());
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
        for (hipStream_t st : {full, masked}) {
            hipEventRecord(a, st);
            hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 100 * 1024, st, 10000ll);     // 100 us at 100 MHz
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s stream: 4096 x 100 us one-per-CU workgroups in %.3f ms (%.1f rounds)\n", st == full ? "full  " : "masked", ms, ms / 0.1);
        }
    // whole-XCD masks (round 6, scripts/gpu_xcd_partition.py): every byte of the mask = the set of XCDs
    for (uint32_t xcds : {0x01u, 0x0fu, 0xf0u, 0x80u}) {
        std::vector<uint32_t> m2(8, xcds * 0x01010101u);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, m2.data()) != hipSuccess) { printf("xcd mask %02x: create failed\n", xcds); continue; }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, st);
            hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 100 * 1024, st, 10000ll);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("XCD mask 0x%02x: 4096 x 100 us one-per-CU workgroups in %.3f ms (%.1f rounds)\n", xcds, ms, ms / 0.1);
        }
    }
    // small-kernel latency under a long big kernel
    for (hipStream_t st : {full, masked}) {
        hipLaunchKernelGGL(spin, dim3(40960), dim3(256), 100 * 1024, st, 10000ll);         // ~16 ms of back-to-back workgroups
        float tot = 0; int n = 0;
        for (int i = 0; i < 20; ++i) {
            hipEventRecord(a, small);
            hipLaunchKernelGGL(spin, dim3(1), dim3(256), 70 * 1024, small, 1000ll);        // 10 us of work, 70 KB LDS
            hipEventRecord(b, small);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); tot += ms; ++n;
        }
        hipStreamSynchronize(st);
        printf("big kernel on the %s stream: a 10-us one-workgroup kernel on another stream takes %.1f us on average\n", st == full ? "full  " : "masked", tot / n * 1e3);
    }
    return 0;
}
