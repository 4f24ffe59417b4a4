// Which physical CUs does bit b of a hipExtStreamCreateWithCUMask mask enable on this stack?  For every bit: a stream with ONLY that
// bit set, a kernel of 64 one-wave workgroups that each record (XCC_ID, HW_ID[15:8] = SE/SH/CU).  Then: two kernels on two
// complementary masked streams run concurrently -- do their workgroups ever share a CU?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ void who(unsigned* out, long long ticks) {
    const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15) << 8) | (hw & 255);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int main() {
    unsigned* d; hipMalloc(&d, 1 << 20);
    std::vector<unsigned> h(1 << 18);
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs: %d\n", ncu);
    std::map<unsigned, int> owner;
    for (int b = 0; b < 256; ++b) {
        std::vector<uint32_t> mask(8, 0u);
        mask[b / 32] = 1u << (b % 32);
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask.data());
        if (e != hipSuccess) { printf("bit %d: create failed %s\n", b, hipGetErrorString(e)); continue; }
        hipLaunchKernelGGL(who, dim3(64), dim3(64), 0, st, d, 200ll);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, 64 * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> ids(h.begin(), h.begin() + 64);
        printf("bit %3d ->", b);
        for (unsigned id : ids) { printf(" xcc %u se/sh/cu 0x%02x", id >> 8, id & 255); owner[id] = b; }
        printf("\n");
        hipStreamDestroy(st);
    }
    printf("distinct CU ids seen: %zu\n", owner.size());
    // complementary masks: "reserved" = the CUs of bits r (r per XCD by the mapping found above is decided by the reader); here:
    // try the two candidate layouts of "2 CUs per XCD" and report how many distinct (xcc) the reserved set covers
    for (int layout = 0; layout < 2; ++layout) {
        std::vector<uint32_t> res(8, 0u), rest(8, 0xFFFFFFFFu);
        for (int x = 0; x < 8; ++x)
            for (int k = 0; k < 2; ++k) {
                const int b = layout == 0 ? x + 8 * k : 32 * x + k;
                res[b / 32] |= 1u << (b % 32);
                rest[b / 32] &= ~(1u << (b % 32));
            }
        hipStream_t s1, s2;
        if (hipExtStreamCreateWithCUMask(&s1, 8, res.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&s2, 8, rest.data()) != hipSuccess) {
            printf("layout %d: create failed\n", layout); continue;
        }
        // big kernel on the complement (4096 workgroups x 256 threads, 50 us each), small kernel on the reserved set meanwhile
        hipLaunchKernelGGL(who, dim3(16384), dim3(256), 0, s2, d, 5000ll);
        hipLaunchKernelGGL(who, dim3(256), dim3(256), 0, s1, d + 65536, 5000ll);
        hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, (65536 + 256) * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> big(h.begin(), h.begin() + 16384), small(h.begin() + 65536, h.begin() + 65536 + 256);
        int shared = 0; for (unsigned id : small) shared += big.count(id);
        std::map<unsigned, int> perx; for (unsigned id : small) perx[id >> 8]++;
        printf("layout %d: complement kernel used %zu CUs, reserved kernel %zu CUs (shared: %d); reserved per xcc:", layout, big.size(), small.size(), shared);
        for (auto& p : perx) printf(" %u:%d", p.first, p.second);
        printf("\n");
        // timing: masked big kernel vs unmasked
        hipStream_t full; hipStreamCreateWithFlags(&full, hipStreamNonBlocking);
        for (hipStream_t st : {full, s2}) {
            hipEventRecord(a, st);
            hipLaunchKernelGGL(who, dim3(16384), dim3(256), 0, st, d, 5000ll);
            hipEventRecord(b2, st); hipEventSynchronize(b2);
            float ms; hipEventElapsedTime(&ms, a, b2);
            printf("   16384 x 50 us workgroups (256 thr) on the %s stream: %.3f ms\n", st == full ? "full" : "complement", ms);
        }
    }
    return 0;
}
