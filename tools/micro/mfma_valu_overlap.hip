// What does non-MFMA work cost next to fp32 MFMA on one SIMD (gfx950)?  Each wave runs ITER iterations of
// MF independent v_mfma_f32_16x16x4_f32 plus M independent instructions of kind OP; time vs M at 4 waves/SIMD.
// Finding (MI355X): fp32 MFMA and VALU do not overlap (the fp32 matrix rate equals the vector rate: same
// FMA lanes), only ~2 VALU per MFMA hide; see DESIGN.md section 5.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_mvo tools/micro/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { OP_FMA = 0, OP_SUB = 1, OP_PKADD = 2, OP_ADDU = 3, OP_BPERM = 4, OP_DPP = 5, OP_LDSREAD = 6, OP_LDSWRITE = 7, OP_LDSWR = 8 };
static const char *kNames[] = {"v_fma_f32", "v_sub_f32", "v_pk_add_f32", "v_add_u32", "ds_bpermute_b32", "v_mov_dpp",
                               "ds_read_b128", "ds_write_b128", "ds_write+read_b128"};

template <int OP, int M, int MF>
__global__ void __launch_bounds__(256) k(float *out, int iters, long long *clk) {
    __shared__ f4 lds[256];
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[16];
    f2 p[8];
    f4 q[4];
    int addr = (threadIdx.x * 4) & 255;
    int waddr = threadIdx.x * 16;
    for (int i = 0; i < 4; i++) q[i] = f4{a, b, a, b};
    lds[threadIdx.x] = f4{a, b, a, b};
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = a + i;
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = f2{a, b};
    __syncthreads();
    long long c0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < MF; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 15]) : "v"(a), "v"(b));
            if (OP == OP_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(b));
            if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(p[(j + 1) & 7]));
            if (OP == OP_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(b));
            if (OP == OP_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(v[j & 15]) : "v"(addr));
            if (OP == OP_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[j & 15]));
            if (OP == OP_LDSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(addr * 4));
            if (OP == OP_LDSWRITE) asm volatile("ds_write_b128 %0, %1" : : "v"(waddr), "v"(q[j & 3]) : "memory");
            if (OP == OP_LDSWR) {
                if (j & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(waddr));
                else asm volatile("ds_write_b128 %0, %1" : : "v"(waddr), "v"(q[(j + 2) & 3]) : "memory");
            }
        }
        if (OP == OP_BPERM || OP >= OP_LDSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long c1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i][0] + p[i][1];
    if (OP >= OP_LDSREAD)
        for (int i = 0; i < 4; i++) s += q[i][0];
    for (int j = 0; j < 4; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int OP, int M, int MF>
void run(int wps) {
    float *out;
    long long *clk, h[2];
    const int blocks = 256 * wps;      // 256-thread blocks: one wave per SIMD; wps blocks per CU
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&clk, 16);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, M, MF><<<blocks, 256>>>(out, 100, clk);
    hipEventRecord(e0);
    k<OP, M, MF><<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ns = ms * 1e6 / iters / wps;   // per wave-iteration per SIMD
    double ghz = (double)h[0] / ((double)h[1] * 10.0);     // wall_clock64 ticks at 100 MHz
    printf("%-16s MFMA %d + %2d ops, %d waves/SIMD: %6.1f ns per wave-iteration  (shader clock %.2f GHz -> %.0f cycles)\n",
           kNames[OP], MF, M, wps, ns, ghz, ns * ghz);
    hipFree(out); hipFree(clk);
}

template <int OP>
void sweep() {
    run<OP, 32, 0>(4);
    run<OP, 8, 4>(4); run<OP, 16, 4>(4); run<OP, 32, 4>(4);
}

int main(int argc, char **) {
    const bool quick = argc > 1;
    run<OP_FMA, 0, 4>(1); run<OP_FMA, 0, 4>(2); run<OP_FMA, 0, 4>(4);
    if (quick) { sweep<OP_LDSREAD>(); sweep<OP_LDSWRITE>(); sweep<OP_LDSWR>(); return 0; }
    sweep<OP_FMA>(); sweep<OP_SUB>(); sweep<OP_PKADD>(); sweep<OP_ADDU>(); sweep<OP_BPERM>(); sweep<OP_DPP>();
    sweep<OP_LDSREAD>(); sweep<OP_LDSWRITE>(); sweep<OP_LDSWR>();
    return 0;
}
