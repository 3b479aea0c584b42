// In which order does v_mfma_f32_32x32x2_f32 add its two products to the accumulator?  Compares one 32x32 tile over
// K = 64 (32 instructions) with CPU fmaf chains in candidate orders: (a) k0 then k1 sequentially, (b) k1 then k0,
// (c) (a0*b0 + a1*b1) rounded once then added.  Operands with wide dynamic range so that orders differ in the last bits.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_mfma32 tools/micro/mfma32_order.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float *A /*[32][64]*/, const float *B /*[32][64]*/, float *D /*[32][32]*/) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f16v acc = {0};
    for (int kk = 0; kk < 64; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 64 + kk + h], B[r * 64 + kk + h], acc, 0, 0, 0);
    // acc[j]: row = 8 * (j / 4) + 4 * h... layout: row = (j / 4) * 8 + h * 4 + (j % 4), col = r
    for (int j = 0; j < 16; ++j) D[((j / 4) * 8 + h * 4 + (j % 4)) * 32 + r] = acc[j];
}
int main() {
    float hA[32 * 64], hB[32 * 64], hD[32 * 32];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        hA[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
        hB[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 12 - 6);
    }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int ok[3] = {0, 0, 0};
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float a = 0, b = 0, c = 0;
            for (int kk = 0; kk < 64; kk += 2) {
                a = fmaf(hA[m * 64 + kk], hB[n * 64 + kk], a); a = fmaf(hA[m * 64 + kk + 1], hB[n * 64 + kk + 1], a);
                b = fmaf(hA[m * 64 + kk + 1], hB[n * 64 + kk + 1], b); b = fmaf(hA[m * 64 + kk], hB[n * 64 + kk], b);
                c = c + fmaf(hA[m * 64 + kk], hB[n * 64 + kk], hA[m * 64 + kk + 1] * hB[n * 64 + kk + 1]);
            }
            ok[0] += (a == hD[m * 32 + n]); ok[1] += (b == hD[m * 32 + n]); ok[2] += (c == hD[m * 32 + n]);
        }
    printf("32x32x2 f32: matches of 1024 outputs -- sequential k0,k1: %d; k1,k0: %d; pair-sum: %d\n", ok[0], ok[1], ok[2]);
    return 0;
}
