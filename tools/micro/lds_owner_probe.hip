// How fast could an LDS-RESIDENT producer of level-1 leaf entries be?  (measured instead of priced.)
// A 256 x 256 fp32 Gram block is 256 KB, the LDS 160 KB: a persistent workgroup per CU owns a row-HALF of one block (128 x 256
// floats = 128 KB).  At 8 codebooks a vector's six level-1 tables read 24 blocks, so every vector and pass makes 48 (block, half)
// items, each needing its own compact lists.  This probe is the CHEAPEST such producer one can write: no position masks, no
// borders, fixed 8 x 8 sub-grids (the lazy tables use about 7.5 x 7.5), lists as 16 contiguous bytes per item, one ds_read per
// lane, 256 bytes out per item, the next item's lists prefetched -- so its time is a floor for the real one.
// Prints ms per 65,536 vectors x 48 items (one pass) for 8 and 16 waves per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_lds_owner tools/micro/lds_owner_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_owner(const float *__restrict__ G, const uint4 *__restrict__ lists, long items,
                                                     float *__restrict__ out) {
    extern __shared__ float half_block[];                // [128][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 128 * 256; e += 64 * WAVES) half_block[e] = G[(size_t)(blockIdx.x % 48) * 128 * 256 + e];
    __syncthreads();
    const long stride = (long)gridDim.x * WAVES;
    long it = (long)blockIdx.x * WAVES + wave;
    if (it >= items) return;
    uint4 cur = lists[it];
    const int i = lane >> 3, j = lane & 7;
    for (; it < items; it += stride) {
        const long nx = it + stride;
        const uint4 nxt = lists[nx < items ? nx : it];           // the next item's lists, a trip ahead
        // bytes 0..7: row entries (< 128), bytes 8..15: column entries (< 256)
        const unsigned ra = (i < 4 ? cur.x >> (8 * i) : cur.y >> (8 * (i - 4))) & 0x7fu;
        const unsigned cb = (j < 4 ? cur.z >> (8 * j) : cur.w >> (8 * (j - 4))) & 0xffu;
        const float v = half_block[ra * 256 + cb];
        __builtin_nontemporal_store(v, out + it * 64 + lane);
        cur = nxt;
    }
}

int main() {
    const long vectors = 65536, items = vectors * 48;
    float *G, *out;
    uint4 *lists;
    hipMalloc(&G, 48ull * 128 * 256 * 4);
    hipMalloc(&out, (size_t)items * 64 * 4);
    hipMalloc(&lists, (size_t)items * 16);
    std::vector<unsigned> h((size_t)items * 4);
    unsigned s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s; }
    hipMemcpy(lists, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(G, 0, 48ull * 128 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, int waves, const char *name) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 256 * 4);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 128 * 256 * 4, 0, G, lists, items, out);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 128 * 256 * 4, 0, G, lists, items, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.3f ms per pass-equivalent (%ld items of 8 x 8 entries, 256 workgroups x %d waves, 128 KB of LDS each): %.1f ns per item and CU\n",
               name, ms / 10, items, waves, ms / 10 * 1e6 / (items / 256.0));
    };
    run(k_owner<8>, 8, "owner, 8 waves per CU");
    run(k_owner<16>, 16, "owner, 16 waves per CU");
    hipError_t e = hipGetLastError();
    printf("last error: %s\n", hipGetErrorString(e));
    return 0;
}
