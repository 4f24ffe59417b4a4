// Phase timing of potrf128_kernel<double> on one random SPD block (developer tool).
#define GPHIP_TIMING 1
#include "../../bayesianinference_amd/csrc/gp_kernels.h"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace gphip;
int main() {
    const int n = 128, ld = 256;
    std::vector<double> A((size_t)ld * ld, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double v = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.1 : 0.0);
        A[(size_t)j * ld + i] = v; A[(size_t)i * ld + j] = v;
    }
    double *dA, *dW, *dP, *dS; int* dI;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, n * n * 8); hipMalloc(&dP, 64); hipMalloc(&dS, 64); hipMalloc(&dI, 4);
    double sp[8] = {1.0, 0.1, 0.0, 1e-14, 0, 0, 0, 0};
    hipMemcpy(dS, sp, 64, hipMemcpyHostToDevice); hipMemset(dI, 0, 4);
    size_t lds = 16 + (size_t)PT_LDS_ELEMS * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(potrf128_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(potrf128_kernel<double>, dim3(1), dim3(256), lds, 0, dA, (long)ld, (long)ld * ld, 0, dW, dP, 1, dI, dS);
        hipDeviceSynchronize();
    }
    long long st[64]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
    auto us = [&](int a, int b) { return (st[b] - st[a]) / 2400.0; };
    printf("load+barrier %.2f us\n", us(0, 1));
    for (int p = 0; p < 8; ++p) {
        if (p < 7) printf("panel %d: update(p-1) || diag %.2f  rowsolve %.2f  (gap %.2f) us\n", p, us(2 + 3 * p, 3 + 3 * p), us(3 + 3 * p, 4 + 3 * p), us(4 + 3 * p, 2 + 3 * (p + 1)));
        else printf("panel 7: diag %.2f us\n", us(23, 24));
    }
    printf("factor total %.2f | L store %.2f | diag inverses %.2f | block inverse %.2f | W store %.2f | total %.2f us\n",
           us(1, 30), us(30, 31), us(31, 32), us(32, 33), us(33, 34), us(0, 34));
    int info; hipMemcpy(&info, dI, 4, hipMemcpyDeviceToHost); printf("info %d\n", info);
    return 0;
}
