// hop_chain.hip -- what a "one wave per work item" kernel costs as a function of its dependent memory round trips:
// 65,536 (or more) single-item waves, each a chain of H dependent loads (index -> table row -> second table ...), at full
// occupancy.  Calibrates the table kernels of the refinement pass (k_tf_er, k_tf_pair*, k_tf_table1, k_tf_comb).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int H, int VW>
__global__ void __launch_bounds__(256) k(const int *__restrict__ idx, const float *__restrict__ T, long n, int tsize, float *__restrict__ out) {
    const long w = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * VW;
    const int lane = threadIdx.x & 63;
    float acc[VW];
    int a[VW];
#pragma unroll
    for (int u = 0; u < VW; ++u) a[u] = (w + u < n) ? idx[(w + u) * 64 + lane] : 0;                     // hop 1: coalesced 256 B
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] = 0.f;
#pragma unroll
    for (int h = 1; h < H; ++h) {
        float v[VW];
#pragma unroll
        for (int u = 0; u < VW; ++u) v[u] = T[(unsigned)a[u] % (unsigned)tsize];                        // hop h + 1: random 4-byte reads from an L2-resident table
#pragma unroll
        for (int u = 0; u < VW; ++u) { acc[u] += v[u]; a[u] = a[u] * 1664525 + (int)(v[u] * 1000.f) + 1013904223; }
    }
#pragma unroll
    for (int u = 0; u < VW; ++u)
        if (w + u < n && lane == 0) out[w + u] = acc[u] + (float)a[u];
}
template <int H, int VW> void run(const int *idx, const float *T, long n, int tsize, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n + 4 * VW - 1) / (4 * VW));
    hipLaunchKernelGGL((k<H, VW>), dim3(grid), dim3(256), 0, 0, idx, T, n, tsize, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<H, VW>), dim3(grid), dim3(256), 0, 0, idx, T, n, tsize, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("items %ld, %d dependent round trips, %d items per wave: %.1f us\n", n, H, VW, ms * 100.f);
}
int main() {
    const long n = 65536;
    const int tsize = 4 << 20;      // 16 MB table
    int *idx; float *T, *out;
    hipMalloc(&idx, n * 64 * 4); hipMalloc(&T, (size_t)tsize * 4); hipMalloc(&out, n * 4);
    hipMemset(idx, 1, n * 64 * 4); hipMemset(T, 0, (size_t)tsize * 4);
    run<1, 1>(idx, T, n, tsize, out);
    run<2, 1>(idx, T, n, tsize, out);
    run<3, 1>(idx, T, n, tsize, out);
    run<4, 1>(idx, T, n, tsize, out);
    run<3, 2>(idx, T, n, tsize, out);
    run<3, 4>(idx, T, n, tsize, out);
    run<2, 4>(idx, T, n, tsize, out);
    return 0;
}
