// HBM WRITE ceiling probe: what a pure 16-byte-per-lane streaming store achieves on this GPU, with the access pattern
// of kbuild_kernel (column-major 128x128 tiles of a leading-dimension-ld matrix, 1 KiB per wave instruction) and as one
// flat stream, plus hipMemsetAsync.  The kernel build writes 4.3 GB of lower-triangle tiles and reads ~nothing, so this
// -- not the 8 TB/s read-side headline -- is the ceiling it runs against.
//   hipcc --offload-arch=gfx950 -O3 -o wbw scripts/micro/wbw.hip && ./wbw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void flat_store(double2* out, long n2) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const double2 v = {1.0 + threadIdx.x, 2.0};
    for (; i < n2; i += stride) out[i] = v;
}
__global__ __launch_bounds__(256) void flat_store_nt(double2* out_, long n2) {
    dv2* out = reinterpret_cast<dv2*>(out_);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const dv2 v = {1.0 + threadIdx.x, 2.0};
    for (; i < n2; i += stride) __builtin_nontemporal_store(v, out + i);
}
// every workgroup owns one contiguous 128 KiB chunk (the memset-like pattern)
__global__ __launch_bounds__(256) void chunk_store(double2* out, int nt_flag) {
    dv2* o = reinterpret_cast<dv2*>(out) + (long)blockIdx.x * 8192 + threadIdx.x;
    const dv2 v = {1.0 + threadIdx.x, 2.0};
    if (nt_flag) { for (int k = 0; k < 32; ++k) __builtin_nontemporal_store(v, o + k * 256); }
    else { for (int k = 0; k < 32; ++k) o[k * 256] = v; }
}
__global__ __launch_bounds__(256) void tile_store_nt(double* out, long ld, int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const double b = 2.0 * nt + 1.0;
    int c = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    while (c > 0 && (long)c * nt - (long)c * (c - 1) / 2 > t) --c;
    while ((long)(c + 1) * nt - (long)(c + 1) * c / 2 <= t) ++c;
    const int tj = c, ti = c + (t - (int)((long)c * nt - (long)c * (c - 1) / 2));
    double* o = out + (long)tj * 128 * ld + (long)ti * 128 + 2 * lane;
    const dv2 v = {1.0 + lane, 2.0 + t};
    for (int jj = wave * 32; jj < wave * 32 + 32; ++jj) __builtin_nontemporal_store(v, reinterpret_cast<dv2*>(o + (long)jj * ld));
}
// one 128x128 fp64 tile per workgroup, lower triangle of an nt x nt tile grid, column-major, ld = nt*128 + 128
__global__ __launch_bounds__(256) void tile_store(double* out, long ld, int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const double b = 2.0 * nt + 1.0;
    int c = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    while (c > 0 && (long)c * nt - (long)c * (c - 1) / 2 > t) --c;
    while ((long)(c + 1) * nt - (long)(c + 1) * c / 2 <= t) ++c;
    const int tj = c, ti = c + (t - (int)((long)c * nt - (long)c * (c - 1) / 2));
    double* o = out + (long)tj * 128 * ld + (long)ti * 128 + 2 * lane;
    const double2 v = {1.0 + lane, 2.0 + t};
    for (int jj = wave * 32; jj < wave * 32 + 32; ++jj) *reinterpret_cast<double2*>(o + (long)jj * ld) = v;
}
int main() {
    const int nt = 256;
    const long ld = (long)nt * 128 + 128;
    const size_t bytes = (size_t)ld * ld * 8;
    double* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const long tiles = (long)nt * (nt + 1) / 2;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(tile_store, dim3((unsigned)tiles), dim3(256), 0, 0, d, ld, nt);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("tile_store (kbuild pattern, %.3f GB): %.3f ms = %.2f TB/s\n", tiles * 131072.0 / 1e9, ms, tiles * 131072.0 / ms / 1e9);
    }
    const long n2 = (long)tiles * 8192;   // same number of bytes, one flat stream
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(flat_store, dim3(8192), dim3(256), 0, 0, reinterpret_cast<double2*>(d), n2);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("flat_store (%.3f GB): %.3f ms = %.2f TB/s\n", n2 * 16.0 / 1e9, ms, n2 * 16.0 / ms / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(flat_store_nt, dim3(8192), dim3(256), 0, 0, reinterpret_cast<double2*>(d), n2);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("flat_store_nt: %.3f ms = %.2f TB/s\n", ms, n2 * 16.0 / ms / 1e9);
    }
    for (int nt_flag = 0; nt_flag < 2; ++nt_flag)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(chunk_store, dim3((unsigned)tiles), dim3(256), 0, 0, reinterpret_cast<double2*>(d), nt_flag);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("chunk_store nt=%d: %.3f ms = %.2f TB/s\n", nt_flag, ms, n2 * 16.0 / ms / 1e9);
        }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(tile_store_nt, dim3((unsigned)tiles), dim3(256), 0, 0, d, ld, nt);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("tile_store_nt: %.3f ms = %.2f TB/s\n", ms, tiles * 131072.0 / ms / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipMemsetAsync(d, 0, (size_t)n2 * 16, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("hipMemsetAsync (%.3f GB): %.3f ms = %.2f TB/s\n", n2 * 16.0 / 1e9, ms, n2 * 16.0 / ms / 1e9);
    }
    hipFree(d);
    return 0;
}
