// LDS write-side rates on gfx950: ds_write_b32/b64/b128 and global_load_lds_dwordx4 (LDS-DMA), 4 waves/SIMD,
// conflict-free lane-linear addresses.  Prints cycles per wave-instruction per CU and bytes/clk/CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_ldsw tools/micro/lds_write_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { W32 = 0, W64 = 1, W128 = 2, DMA16 = 3, DMA4 = 4, R128 = 5 };
static const char *kNames[] = {"ds_write_b32", "ds_write_b64", "ds_write_b128", "global_load_lds_dwordx4", "global_load_lds_dword",
                               "ds_read_b128"};
static const int kBytes[] = {4, 8, 16, 16, 4, 16};

template <int OP, int M>
__global__ void __launch_bounds__(256) k(const float *src, float *out, int iters, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f4 q = {1.f, 2.f, 3.f, 4.f};
    f4 rd[4];
    const unsigned wbase = wave * 4096;             // 4 KB region per wave
    const unsigned a32 = wbase + lane * 4, a64 = wbase + lane * 8, a128 = wbase + lane * 16;
    const char *g = reinterpret_cast<const char *>(src) + ((blockIdx.x & 63) * 256 + threadIdx.x) * 16;
    long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            const unsigned o = (j & 3) * 1024;
            if (OP == W32) asm volatile("ds_write_b32 %0, %1" : : "v"(a32 + o), "v"(q[0]) : "memory");
            if (OP == W64) asm volatile("ds_write_b64 %0, %1" : : "v"(a64 + o), "v"(*reinterpret_cast<f2 *>(&q)) : "memory");
            if (OP == W128) asm volatile("ds_write_b128 %0, %1" : : "v"(a128 + o), "v"(q) : "memory");
            if (OP == R128) asm volatile("ds_read_b128 %0, %1" : "=v"(rd[j & 3]) : "v"(a128 + o));
            if (OP == DMA16 || OP == DMA4) {
                unsigned keep;
                const unsigned dst = wbase + o;
                if (OP == DMA16)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
                else
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
            }
        }
        if (OP == DMA16 || OP == DMA4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    __syncthreads();
    float s = reinterpret_cast<float *>(smem)[threadIdx.x];
    if (OP == R128) for (int i = 0; i < 4; i++) s += rd[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int OP, int M>
void run() {
    const int wps = 4, blocks = 256 * wps, iters = 4000;
    float *out, *src; long long *clk, h[2];
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&src, 64 * 256 * 16);
    (void)hipMemset(src, 0, 64 * 256 * 16);
    (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP, M><<<blocks, 256, 16384>>>(src, out, 50, clk);
    (void)hipEventRecord(e0);
    k<OP, M><<<blocks, 256, 16384>>>(src, out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    // per CU: 16 waves each issue M instructions per iteration
    const double cyc_per_instr_cu = ms * 1e6 * ghz / iters / (16.0 * M);
    printf("%-26s x%2d/iter: %.2f cycles per wave-instruction per CU  = %.0f B/clk/CU  (clock %.2f GHz)\n", kNames[OP], M,
           cyc_per_instr_cu, 64.0 * kBytes[OP] / cyc_per_instr_cu, ghz);
    (void)hipFree(out); (void)hipFree(src); (void)hipFree(clk);
}

int main() {
    run<R128, 16>(); run<W32, 16>(); run<W64, 16>(); run<W128, 16>(); run<DMA16, 4>(); run<DMA16, 16>(); run<DMA4, 16>();
    return 0;
}
