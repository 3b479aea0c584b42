// Dependent v_mfma_f32_16x16x4_f32 chains on gfx950: cycles per MFMA with NCH independent accumulators per wave
// at 1/2/4 waves per SIMD (a bit-exact k-ordered dot product is ONE chain per output tile).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_chain tools/micro/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ void __launch_bounds__(256) k(float *out, int iters, long long *clk) {
    f4 acc[NCH];
    for (int j = 0; j < NCH; j++) acc[j] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % NCH], 0, 0, 0);
    }
    long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int j = 0; j < NCH; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NCH>
void run(int wps) {
    const int blocks = 256 * wps, iters = 20000;
    float *out; long long *clk, h[2];
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NCH><<<blocks, 256>>>(out, 100, clk);
    (void)hipEventRecord(e0);
    k<NCH><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    printf("chains/wave %d, waves/SIMD %d: %.1f cycles per MFMA per SIMD (clock %.2f GHz)\n", NCH, wps,
           ms * 1e6 * ghz / iters / 8.0 / wps, ghz);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    for (int wps : {1, 2, 3, 4, 8}) { run<1>(wps); run<2>(wps); run<4>(wps); }
    return 0;
}
