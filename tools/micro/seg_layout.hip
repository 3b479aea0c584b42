// Stage 0 reads, per (vector, codebook n) wave, 7 pseudo-random 1 KB row segments of the Gram matrix; workgroup id mod 8 = n,
// so XCD n only touches column segment n.  Does it matter to the L2 whether that segment is laid out as it is today (row r of
// segment n at r * 8 KB + n * 1 KB: the XCD's 2 MB working set is 2,048 pieces strided by 8 KB) or segment-major (n * 2 MB +
// r * 1 KB: contiguous)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_seg tools/micro/seg_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MAJOR>
__global__ void __launch_bounds__(256) k(const char *G, float *out, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const unsigned n = blockIdx.x & 7;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned h = wid * 2654435761u + 12345u;
    f4 acc = {0, 0, 0, 0};
    f4 v[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        h = h * 1664525u + 1013904223u;
        const unsigned row = (h >> 10) & 2047u;
        const size_t off = MAJOR ? ((size_t)n * 2048 + row) * 1024 : (size_t)row * 8192 + n * 1024;
        v[u] = *reinterpret_cast<const f4 *>(G + off + lane * 16);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) acc += v[u];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    char *G; float *out;
    (void)hipMalloc(&G, 16 << 20);
    (void)hipMemset(G, 0, 16 << 20);
    const unsigned blocks = 65536u * 8 / 4;            // 65,536 vectors x 8 codebooks, four waves per workgroup
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int major = 0; major < 2; ++major)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (major) k<1><<<blocks, 256>>>(G, out, 7); else k<0><<<blocks, 256>>>(G, out, 7);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms  %.1f TB/s of row segments\n", major ? "segment-major" : "row-major (today)", ms, blocks * 4.0 * 7 * 1024 / ms / 1e9);
        }
    return 0;
}
