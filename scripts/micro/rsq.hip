// accuracy of v_rsq_f64 + one Newton step vs correctly rounded 1/sqrt (host long double)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* l1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double d = x[i];
    double y = __builtin_amdgcn_rsq(d);
    y0[i] = y;
    double t = d * y, h = 0.5 * y;
    double e = __builtin_fma(-t, h, 0.5);
    y1[i] = __builtin_fma(y, e, y);
    l1[i] = __builtin_fma(t, e, t);
}
int main() {
    const int n = 1 << 16; std::vector<double> x(n), a(n), b(n), c(n);
    for (int i = 0; i < n; ++i) x[i] = std::exp(-30.0 + 60.0 * (i + 0.5) / n) * (1.0 + 0.37 * std::sin(i));
    double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double r = 1.0L / sqrtl((long double)x[i]), s = sqrtl((long double)x[i]);
        e0 = fmax(e0, (double)fabsl((a[i] - r) / r)); e1 = fmax(e1, (double)fabsl((b[i] - r) / r)); e2 = fmax(e2, (double)fabsl((c[i] - s) / s));
    }
    printf("max rel err: raw v_rsq_f64 %.3e | +1 Newton (rsqrt) %.3e | +1 Newton (sqrt) %.3e\n", e0, e1, e2);
    return 0;
}
