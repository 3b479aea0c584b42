// L2 -> L1 gather bandwidth on gfx950: every wave-load fetches 1 KB as 64 / PIECE contiguous pieces of PIECE bytes taken
// from pseudo-random rows of an L2-resident table (4 MB).  Does the piece size (64 B = half a cache line, 128, 256 B) matter?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_l2g tools/micro/l2_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PIECE>
__global__ void __launch_bounds__(256) k(const char *tab, unsigned rows_mask, int row_bytes, float *out, int iters) {
    constexpr int LPP = PIECE / 16;                       // lanes per piece
    const int lane = threadIdx.x & 63;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned h = wid * 2654435761u + 12345u;
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += 4) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned piece = lane / LPP;            // which piece of this wave-load
            const unsigned r = (h ^ (piece * 0x9E3779B9u) ^ ((piece * 7919u) << 7)) >> 8;
            const unsigned row = r & rows_mask;
            const unsigned col = ((r >> 20) * PIECE) % (unsigned)row_bytes;   // piece-aligned column
            v[u] = *reinterpret_cast<const f4 *>(tab + (size_t)row * row_bytes + col + (lane % LPP) * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int PIECE>
void run(const char *tab, float *out, size_t table_bytes) {
    const int row_bytes = 2048, iters = 2048, blocks = 256 * 8;   // 8192 waves x 2048 loads x 1 KB = 17 GB
    const unsigned rows_mask = (unsigned)(table_bytes / row_bytes) - 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PIECE><<<blocks, 256>>>(tab, rows_mask, row_bytes, out, 64);
    (void)hipEventRecord(e0);
    k<PIECE><<<blocks, 256>>>(tab, rows_mask, row_bytes, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 1024.0;
    printf("table %4zu KB, pieces of %3d B: %.2f ms  %.1f TB/s into registers\n", table_bytes >> 10, PIECE, ms, bytes / ms / 1e9);
}

int main() {
    char *tab; float *out;
    (void)hipMalloc(&tab, 64 << 20);
    (void)hipMemset(tab, 0, 64 << 20);
    (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (size_t tb : {(size_t)512 << 10, (size_t)4 << 20, (size_t)32 << 20}) {
        run<64>(tab, out, tb); run<128>(tab, out, tb); run<256>(tab, out, tb); run<1024>(tab, out, tb);
    }
    return 0;
}
