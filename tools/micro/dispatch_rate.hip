// dispatch_rate.hip -- how fast the chip replaces finished single-wave workgroups, as a function of the workgroup's LDS and of
// how long a wave lives.  The pass kernels of the refinement launch 262,144 ... 524,288 one-wave workgroups per launch; round 5's
// lean level-1 tables made the waves live 11 % shorter and the launch no faster -- the CUs held fewer waves.  Is the dispatcher
// the limit?  Kernel: one wave per workgroup, LDS bytes static, body = SPIN dependent s_memtime reads (~ clocks of life).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void __launch_bounds__(64) k(float *__restrict__ out, int spin) {
    __shared__ float buf[LDS / 4 > 0 ? LDS / 4 : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while ((long long)(t - t0) < spin) t = __builtin_amdgcn_s_memtime();
    if (LDS > 0) buf[threadIdx.x] = (float)t;
    if (t == 1 && out) out[blockIdx.x] = LDS > 0 ? buf[(threadIdx.x + 1) & 63] : 0.f;      // (never: keeps the LDS array)
}
template <int LDS> void run(long wgs, int spin, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LDS>), dim3((unsigned)wgs), dim3(64), 0, 0, out, spin);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<LDS>), dim3((unsigned)wgs), dim3(64), 0, 0, out, spin);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5;
    // with W waves resident per CU and a life of `spin` clocks, the floor is wgs / (256 CU * W) * spin clocks
    printf("LDS %5d B, %7ld one-wave workgroups, life >= %5d ticks: %8.1f us  (%.2f workgroups per ns; per CU one start per %.0f ns)\n", LDS, wgs, spin,
           us, wgs / us / 1e3, us * 1e3 * 256 / wgs);
}
int main() {
    float *out; hipMalloc(&out, 1 << 22);
    for (long wgs : {262144L, 393216L}) {
        for (int spin : {0, 1000, 3000, 6000, 12000}) {
            run<0>(wgs, spin, out);
            run<1664>(wgs, spin, out);
            run<5056>(wgs, spin, out);
        }
    }
    return 0;
}
