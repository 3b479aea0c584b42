// i8_peak.hip -- sustained rate of v_mfma_i32_32x32x32_i8 with the accumulator pattern of k_fgemm (16 accumulators of
// 16 registers, 40 MFMAs per "k step"), one wave per SIMD, no memory traffic: the ceiling the GEMM's main loop can reach.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_peak(int iters, int *out) {
    v16i acc[16];
    for (int i = 0; i < 16; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0;
    v4i a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = (v4i){(int)threadIdx.x * 3 + i, i, 7 * i, (int)threadIdx.x};
        b[i] = (v4i){(int)threadIdx.x * 5 + i, 2 * i, 3 * i, (int)threadIdx.x ^ i};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 40; ++q) acc[q & 15] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q & 7], b[(q >> 1) & 7], acc[q & 15], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 16; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    if (s == 123456789) out[0] = s;
}
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int wgs = argc > 2 ? atoi(argv[2]) : 256;
    int *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak, dim3(wgs), dim3(256), 0, 0, 100, d);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak, dim3(wgs), dim3(256), 0, 0, iters, d);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)wgs * 4 * iters * 40 * 65536.0;
        printf("iters %d wgs %d: %.3f ms, %.0f TOPS; per k step of 40 MFMAs: %.3f us\n", iters, wgs, ms, ops / ms * 1e-9, ms * 1e3 / iters);
    }
    return 0;
}
