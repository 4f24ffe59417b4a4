// Does a stream-ordered wait on a device word work here, and what does it cost?  (Round 6: the prerequisite of "tile column
// final" signals from inside a dataflow panel launch of the sharded schedule.)  Stream A runs a kernel that spins ~1 ms, bumps a
// counter in signal memory, spins another ~1 ms; stream B waits for the counter (hipStreamWaitValue32) and then runs a kernel that
// stamps the time.  Expected: B's stamp lands ~at A's first milestone, not at A's end.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void producer(unsigned int* sig, long long* t, long long spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    t[0] = wall_clock64();
    __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    while (wall_clock64() - t0 < 2 * spin) {}
    t[1] = wall_clock64();
}
__global__ void consumer(long long* t) { t[2] = wall_clock64(); }
int main() {
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    long long* t; hipMalloc(&t, 64);
    for (int kind = 0; kind < 2; ++kind) {
        unsigned int* sig = nullptr;
        hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) : hipMalloc((void**)&sig, 8);
        if (e != hipSuccess) { printf("alloc kind %d failed: %s\n", kind, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemset(sig, 0, 8); hipMemset(t, 0, 64);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, a, sig, t, 100000ll);      // 100 MHz clock: 1 ms
        e = hipStreamWaitValue32(b, sig, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
        if (e != hipSuccess) { printf("%s memory: hipStreamWaitValue32 -> %s\n", kind ? "plain" : "signal", hipGetErrorString(e)); (void)hipGetLastError(); hipDeviceSynchronize(); continue; }
        hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, t);
        hipDeviceSynchronize();
        long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
        printf("%s memory: consumer ran %.1f us after the signal, %.1f us before the producer ended\n", kind ? "plain" : "signal",
               (h[2] - h[0]) / 100.0, (h[1] - h[2]) / 100.0);
        hipFree(sig);
    }
    return 0;
}
