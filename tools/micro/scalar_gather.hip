// Can the scalar unit take part of a table kernel's gathers?  k_tf_level1 is bound by the vector L1's access rate (one access per
// 16-lane group and 64-byte piece); scalar loads reach the L2 through the scalar cache instead.  Rate of random 4-byte scalar loads
// from a 16 MB table (no reuse: every load is a scalar-cache miss), alone and beside vector gathers of the level-1 kind.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_sg tools/micro/scalar_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NS, int NV>
__global__ void __launch_bounds__(64) k(const float *G, float *out) {
    const unsigned wid = blockIdx.x;
    const int lane = threadIdx.x;
    unsigned h = wid * 2654435761u + 12345u;
    float acc = 0.f;
    // NV vector gathers: 64 lanes, random 4-byte entries (every lane its own 64-byte piece: 64 L1 accesses each)
    float vg[NV > 0 ? NV : 1];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        unsigned hv = (h + u * 977u + lane * 7919u) * 1664525u + 1013904223u;
        vg[u] = G[(hv >> 8) & ((4u << 20) - 1u)];
    }
    // NS scalar loads, all requested before the first is used (batches of 8)
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += 8) {
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned idx = __builtin_amdgcn_readfirstlane((h >> 8) & ((4u << 20) - 1u));
            sv[u] = G[idx];      // wave-uniform address: a scalar load
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += sv[u];
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) acc += vg[u];
    if (acc == 12345.678f) out[wid] = acc;      // (never: keeps the loads)
    if (lane == 0) out[wid] = acc;
}

int main() {
    float *G, *out;
    (void)hipMalloc(&G, 16 << 20); (void)hipMemset(G, 0, 16 << 20);
    const unsigned waves = 65536u * 6;
    (void)hipMalloc(&out, (size_t)waves * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define RUN(NS, NV)                                                                                         \
    for (int rep = 0; rep < 3; ++rep) {                                                                     \
        (void)hipEventRecord(e0);                                                                           \
        k<NS, NV><<<waves, 64>>>(G, out);                                                                   \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                            \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                   \
        if (rep == 2) printf("%3d scalar loads + %d vector gathers (64 accesses each) per wave, %u waves: %.3f ms\n", NS, NV, waves, ms); \
    }
    RUN(0, 4) RUN(0, 5) RUN(0, 6)
    RUN(16, 0) RUN(32, 0) RUN(64, 0) RUN(128, 0)
    RUN(16, 4) RUN(32, 4) RUN(64, 4) RUN(64, 5)
    return 0;
}
