// cmp_rates.hip -- issue cost of the selection's inner operations: rank += (a < b) with 64-bit vs 32-bit keys, and
// v_readlane pairs; 4 waves per SIMD, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, u64 *out, u64 seed) {
    u64 k0 = seed * (threadIdx.x + 1) + blockIdx.x;
    int r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 o[8];
    for (int u = 0; u < 8; ++u) o[u] = (seed ^ 0x9e3779b97f4a7c15ull) * (u + 3 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            o[u] ^= (u64)(unsigned)(it + u) << 20;                  // one VALU op on a lane-varying stream
            if (MODE == 0) r[u] += (o[u] < k0) ? 1 : 0;
            if (MODE == 1) r[u] += ((unsigned)o[u] < (unsigned)k0) ? 1 : 0;
            if (MODE == 2) r[u] += ((unsigned)(o[u] >> 32) < (unsigned)(k0 >> 32) || ((unsigned)(o[u] >> 32) == (unsigned)(k0 >> 32) && (unsigned)o[u] < (unsigned)k0)) ? 1 : 0;
            if (MODE == 3) r[u] += 1;
        }
    }
    int s = 0;
    for (int u = 0; u < 8; ++u) s += r[u] + (int)(o[u] >> 40);
    if (s == 0x7fffffff) out[0] = s;
}
template <int MODE> void run(const char *name) {
    u64 *d; hipMalloc(&d, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, 10, d, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, iters, d, 12345ull);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 8 rank updates
    printf("%-28s %.3f ms: %.2f ns per rank update per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / (4.0 * iters * 8), ms * 1e6 / (4.0 * iters * 8) * 2.4);
}
int main() {
    run<0>("u64 key  (o < k)");
    run<1>("u32 key");
    run<2>("u64 as two u32 compares");
    run<3>("no compare (xor + add only)");
    return 0;
}
