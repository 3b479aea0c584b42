// Gram-table lookups on gfx950: how fast can a wave collect the 17 x 17 sub-matrix G[R_n][R_m] (R = 16 shortlisted
// entries + the current one, pseudo-random) of an L2-resident fp32 Gram matrix G = C C^T ([NK][NK], NK = 2048: 16 MB)?
//   V1: four global_load_dword gathers per lane (16 x 16 core) + one for the 33 border values
//   V2: the 17 row segments (256 floats = 1 KB each) as coalesced float4 loads into LDS, then ds_read_b32 picks
// Workgroup = one wave per (vector, pair table), pair = blockIdx % npairs so that an XCD only touches its own tables.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_gram tools/micro/gram_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NK = 2048, K = 256;

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V>
__global__ void __launch_bounds__(64) k(const float *G, const int2 *pairs, int npairs, float *out) {
    __shared__ float seg[17 * 256];
    const int lane = threadIdx.x;
    const int p = blockIdx.x % npairs;
    const unsigned b = blockIdx.x / npairs;
    const int2 nm = pairs[p];
    // 17 pseudo-random row / column entries of this (vector, codebook)
    const unsigned hr = hash(b * 64u + nm.x), hc = hash(b * 64u + nm.y);
    auto ent = [&](unsigned h, int i) { return (int)(hash(h + i * 0x9E3779B9u) & (K - 1)); };
    float acc = 0.f;
    if (V == 3) {
        // as V1 with the 16 column entries in ASCENDING order (one per block of 16 columns, pseudo-random inside it): the four
        // lanes of a quad read neighbouring columns of one row -- what entry-ordered shortlists would give the leaf tables
        const int j = lane & 15;
        const int col = nm.y * K + 16 * j + (int)(hash(hc + j) & 15);
        float v[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = nm.x * K + ent(hr, 4 * q + (lane >> 4));
            v[q] = G[(size_t)row * NK + col];
        }
        {
            const int row = nm.x * K + ent(hr, lane < 16 ? lane : 16);
            const int colb = nm.y * K + ent(hc, lane < 16 ? 16 : (lane < 32 ? lane - 16 : 16));
            v[4] = G[(size_t)row * NK + colb];
        }
        acc = ((v[0] + v[1]) + (v[2] + v[3])) + v[4];
    } else if (V == 1) {
        const int j = lane & 15;
        const int col = nm.y * K + ent(hc, j);
        float v[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = nm.x * K + ent(hr, 4 * q + (lane >> 4));
            v[q] = G[(size_t)row * NK + col];
        }
        {   // borders: lanes 0..15 (row i, old col), 16..31 (old row, col j), 32 (old, old)
            const int row = nm.x * K + ent(hr, lane < 16 ? lane : 16);
            const int colb = nm.y * K + ent(hc, lane < 16 ? 16 : (lane < 32 ? lane - 16 : 16));
            v[4] = G[(size_t)row * NK + colb];
        }
        acc = ((v[0] + v[1]) + (v[2] + v[3])) + v[4];
    } else {
        for (int i = 0; i < 17; ++i) {
            const int row = nm.x * K + ent(hr, i);
            const f4 t = *reinterpret_cast<const f4 *>(G + (size_t)row * NK + nm.y * K + 4 * lane);
            *reinterpret_cast<f4 *>(seg + i * 256 + 4 * lane) = t;
        }
        __syncthreads();
        const int cj = ent(hc, lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += seg[(4 * q + (lane >> 4)) * 256 + cj];
        acc += seg[(lane < 17 ? lane : 0) * 256 + ent(hc, 16)] + seg[16 * 256 + cj];
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;      // the 1 KB D table of this (vector, pair)
}

int main() {
    float *G, *out; int2 *dp;
    (void)hipMalloc(&G, (size_t)NK * NK * 4);
    (void)hipMemset(G, 0, (size_t)NK * NK * 4);
    const unsigned B = 65536;
    (void)hipMalloc(&out, (size_t)B * 28 * 64 * 4);
    (void)hipMalloc(&dp, 64 * sizeof(int2));
    struct Cfg { const char *name; std::vector<int2> pairs; };
    std::vector<Cfg> cfgs;
    cfgs.push_back({"L1 (4 sibling tables)", {{0, 1}, {2, 3}, {4, 5}, {6, 7}}});
    cfgs.push_back({"L2 (8 tables)", {{0, 2}, {4, 6}, {0, 3}, {4, 7}, {1, 2}, {5, 6}, {1, 3}, {5, 7}}});
    { Cfg c{"L4 (16 tables)", {}}; for (int a = 0; a < 4; ++a) for (int b = 4; b < 8; ++b) c.pairs.push_back({a, b}); cfgs.push_back(c); }
    { Cfg c{"all 28", {}}; for (int a = 0; a < 8; ++a) for (int b = a + 1; b < 8; ++b) c.pairs.push_back({a, b}); cfgs.push_back(c); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (auto &c : cfgs) {
        const int np = (int)c.pairs.size();
        (void)hipMemcpy(dp, c.pairs.data(), np * sizeof(int2), hipMemcpyHostToDevice);
        for (int v = 1; v <= 3; ++v) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (v == 1) k<1><<<B * np, 64>>>(G, dp, np, out); else if (v == 2) k<2><<<B * np, 64>>>(G, dp, np, out); else k<3><<<B * np, 64>>>(G, dp, np, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("%-24s V%d: %.3f ms for %u vectors x %d tables = %.1f G lookups/s (289 per table), %.2f us per 1000 tables\n",
                   c.name, v, best, B, np, 289.0 * B * np / best / 1e6, best * 1e3 / (B * (double)np / 1000));
        }
    }
    return 0;
}
