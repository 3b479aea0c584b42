// What separates k_tf_stage0<256,8> without its selection (0.226 ms) from its access pattern alone (seg_layout.hip: 0.13 ms)?
// The pattern with the kernel's other ingredients added one at a time:
//   IDX  the seven row numbers come from the vector's 8 index bytes (one scalar 8-byte load the row addresses depend on)
//   XC   the wave's 1 KB x.C segment, nontemporal, from a 0.5 GB array (HBM)
//   Q    the codebook's 1 KB of |c|^2 (L1 / L2 resident)
//   OUT  16 entries + 16 scores per wave (80 bytes) instead of 1 KB
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_s0i tools/micro/stage0_ingredients.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../quantization_amd/csrc/mcq_kernels.h"
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool IDX, int XC, bool Q, bool OUT, int SEL = 0, int PF = 0>
__global__ void __launch_bounds__(256) k(const char *G, const unsigned long long *idx, const float *xc, const float *q, float *out, uint8_t *ent, float *S) {
    const int lane = threadIdx.x & 63;
    const unsigned n = blockIdx.x & 7;
    const unsigned b = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    f4 xv = {0, 0, 0, 0}, qv = {0, 0, 0, 0};
    // XC: 1 nontemporal from the 0.5 GB array (HBM), 2 the same with a plain load, 3 from a 16 MB window of it (cache resident)
    if (XC == 1) xv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xc + ((size_t)b * 2048 + n * 256 + 4 * lane)));
    if (XC == 2) xv = *reinterpret_cast<const f4 *>(xc + ((size_t)b * 2048 + n * 256 + 4 * lane));
    if (XC == 10) xv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xc + (((size_t)n * 65536 + b) * 256 + 4 * lane)));      // codebook-major x.C: an XCD streams ONE contiguous 64 MB region
    if (XC == 9) xv = *reinterpret_cast<const f4 *>(xc + ((size_t)(b & 2047) * 2048 + n * 256 + 4 * lane));      // 16 MB window, plain load: L2 hits
    if (XC >= 3 && XC < 9) xv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xc + ((size_t)(b & ((1u << XC) * 256u - 1u)) * 2048 + n * 256 + 4 * lane)));      // window of 2^XC * 2 MB
    if (Q) qv = *reinterpret_cast<const f4 *>(q + n * 256 + 4 * lane);
    // PF: the x.C segment of the vector PF places further on (same codebook = same XCD) is pulled into THIS XCD's L2 through the
    // scalar unit -- one 4-byte scalar load per 128-byte line, nobody waits for them before the end of the wave
    int pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (PF > 0) {
        const unsigned bp = b + PF < 65536u ? b + PF : b;
        const float *pp = xc + ((size_t)__builtin_amdgcn_readfirstlane((int)bp) * 2048 + n * 256);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("s_load_dword %0, %1, %2" : "=s"(pf[i]) : "s"(pp), "i"(i * 128));
    }
    unsigned long long iw = 0;
    if (IDX) iw = idx[__builtin_amdgcn_readfirstlane((int)b)];
    unsigned h = wid * 2654435761u + 12345u;
    f4 v[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        h = h * 1664525u + 1013904223u;
        const unsigned m = u < n ? u : u + 1;
        const unsigned row = IDX ? m * 256 + (unsigned)((iw >> (8 * m)) & 0xff) : (h >> 10) & 2047u;
        v[u] = *reinterpret_cast<const f4 *>(G + (size_t)row * 8192 + n * 1024 + lane * 16);
    }
    f4 acc = v[0];
#pragma unroll
    for (int u = 1; u < 7; ++u) acc += v[u];
    acc = (qv + 2.0f * (acc - xv));
    if (SEL >= 2) {      // a wave that merely stays on for SEL dependent steps after its loads (no LDS, no scalar loop)
        float t = acc[0];
#pragma unroll 1
        for (int i = 0; i < SEL; ++i) t = __builtin_amdgcn_readlane(t, (i * 7) & 63) * 1.0001f + acc[1];
        if (lane < 16) { ent[((size_t)b * 8 + n) * 16 + lane] = (uint8_t)lane; S[((size_t)b * 8 + n) * 16 + lane] = t; }
        return;
    }
    if (SEL == 1) {      // the kernel's own selection: the 16 smallest of the wave's 256 scores, listed in ascending entry
        __shared__ mcq::u64 sel[4][mcq::kSelectLdsU64];
        float sv[4] = {acc[0], acc[1], acc[2], acc[3]};
        int sp[4] = {4 * lane, 4 * lane + 1, 4 * lane + 2, 4 * lane + 3};
        float ov; int op, dst; bool has;
        mcq::wave_select_set<4>(sv, sp, 16, 256, sel[threadIdx.x >> 6], has, dst, ov, op);
        if (has) { ent[((size_t)b * 8 + n) * 16 + dst] = (uint8_t)op; S[((size_t)b * 8 + n) * 16 + dst] = ov; }
        if (PF > 0) asm volatile("" :: "s"(pf[0]), "s"(pf[1]), "s"(pf[2]), "s"(pf[3]), "s"(pf[4]), "s"(pf[5]), "s"(pf[6]), "s"(pf[7]));
        return;
    }
    if (OUT) {
        if (lane < 16) { ent[((size_t)b * 8 + n) * 16 + lane] = (uint8_t)lane; S[((size_t)b * 8 + n) * 16 + lane] = acc[0] + acc[1] + acc[2] + acc[3]; }
    } else out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the same work with a wave walking V consecutive vectors of its codebook and its x.C segment requested AHEAD iterations early
// (after the rows of the current vector: loads return in order, so the wait for the rows must not sit behind an HBM load)
template <int V, int AHEAD>
__global__ void __launch_bounds__(256) kl(const char *G, const unsigned long long *idx, const float *xc, const float *q, uint8_t *ent, float *S) {
    __shared__ mcq::u64 sel[4][mcq::kSelectLdsU64];
    const int lane = threadIdx.x & 63;
    const unsigned n = blockIdx.x & 7;
    const unsigned b0 = ((blockIdx.x >> 3) * 4 + (threadIdx.x >> 6)) * V;
    const f4 qv = *reinterpret_cast<const f4 *>(q + n * 256 + 4 * lane);
    auto xload = [&](unsigned b) { return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xc + ((size_t)b * 2048 + n * 256 + 4 * lane))); };
    f4 xq[AHEAD + 1];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) xq[a] = xload(b0 + (a < V ? a : V - 1));
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const unsigned b = b0 + v;
        const unsigned long long iw = idx[__builtin_amdgcn_readfirstlane((int)b)];
        f4 r[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const unsigned m = u < n ? u : u + 1;
            const unsigned row = m * 256 + (unsigned)((iw >> (8 * m)) & 0xff);
            r[u] = *reinterpret_cast<const f4 *>(G + (size_t)row * 8192 + n * 1024 + lane * 16);
        }
        if (AHEAD > 0) xq[AHEAD] = xload(b0 + (v + AHEAD < V ? v + AHEAD : V - 1));
        else xq[0] = xload(b);
        f4 acc = r[0];
#pragma unroll
        for (int u = 1; u < 7; ++u) acc += r[u];
        acc = (qv + 2.0f * (acc - xq[0]));
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) xq[a] = xq[a + 1];
        float sv[4] = {acc[0], acc[1], acc[2], acc[3]};
        int sp[4] = {4 * lane, 4 * lane + 1, 4 * lane + 2, 4 * lane + 3};
        float ov; int op, dst; bool has;
        mcq::wave_select_set<4>(sv, sp, 16, 256, sel[threadIdx.x >> 6], has, dst, ov, op);
        if (has) { ent[((size_t)b * 8 + n) * 16 + dst] = (uint8_t)op; S[((size_t)b * 8 + n) * 16 + dst] = ov; }
    }
}

// Variants of HOW the x.C segment reaches the wave (everything else as "+ selection (everything)"):
//   MODE 0  plain: requested first (the shipped order)          MODE 1  requested AFTER the rows
//   MODE 2  by LDS-DMA into the wave's own 1 KB of LDS          MODE 3  by a FIFTH wave of the workgroup for the four vectors (LDS, barrier)
template <int MODE>
__global__ void __launch_bounds__(MODE == 3 ? 320 : 256) kx(const char *G, const unsigned long long *idx, const float *xc, const float *q, uint8_t *ent, float *S) {
    __shared__ mcq::u64 sel[4][mcq::kSelectLdsU64];
    __shared__ __attribute__((aligned(16))) float xl[4][256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned n = blockIdx.x & 7;
    if (MODE == 3 && wave == 4) {      // the loader: four 1 KB segments, then the barrier the row waves meet it at
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned b = (blockIdx.x >> 3) * 4 + w;
            const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xc + ((size_t)b * 2048 + n * 256 + 4 * lane)));
            *reinterpret_cast<f4 *>(&xl[w][4 * lane]) = v;
        }
        __syncthreads();
        return;
    }
    const unsigned b = (blockIdx.x >> 3) * 4 + wave;
    f4 xv = {0, 0, 0, 0};
    const float *xp = xc + ((size_t)b * 2048 + n * 256 + 4 * lane);
    if (MODE == 0) xv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xp));
    if (MODE == 2) {
        const unsigned d = (unsigned)(size_t)&xl[wave][0];
        const float *base = xc + ((size_t)b * 2048 + n * 256);
        const unsigned vo = lane * 16;
        asm volatile("s_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[p]" : : [vo] "v"(vo), [p] "s"(base), [d] "s"(d) : "memory");
    }
    const f4 qv = *reinterpret_cast<const f4 *>(q + n * 256 + 4 * lane);
    const unsigned long long iw = idx[__builtin_amdgcn_readfirstlane((int)b)];
    f4 v[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const unsigned m = u < n ? u : u + 1;
        const unsigned row = m * 256 + (unsigned)((iw >> (8 * m)) & 0xff);
        v[u] = *reinterpret_cast<const f4 *>(G + (size_t)row * 8192 + n * 1024 + lane * 16);
    }
    if (MODE == 1) xv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xp));
    f4 acc = v[0];
#pragma unroll
    for (int u = 1; u < 7; ++u) acc += v[u];
    if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); xv = *reinterpret_cast<const f4 *>(&xl[wave][4 * lane]); }
    if (MODE == 3) { __syncthreads(); xv = *reinterpret_cast<const f4 *>(&xl[wave][4 * lane]); }
    acc = (qv + 2.0f * (acc - xv));
    float sv[4] = {acc[0], acc[1], acc[2], acc[3]};
    int sp[4] = {4 * lane, 4 * lane + 1, 4 * lane + 2, 4 * lane + 3};
    float ov; int op, dst; bool has;
    mcq::wave_select_set<4>(sv, sp, 16, 256, sel[wave], has, dst, ov, op);
    if (has) { ent[((size_t)b * 8 + n) * 16 + dst] = (uint8_t)op; S[((size_t)b * 8 + n) * 16 + dst] = ov; }
}

int main() {
    const unsigned B = 65536;
    char *G; float *out, *xc, *q, *S; unsigned long long *idx; uint8_t *ent;
    (void)hipMalloc(&G, 16 << 20);
    { float *h = (float *)malloc(16 << 20); for (int i = 0; i < (4 << 20); ++i) h[i] = (float)(rand() & 0xffff) / 65536.0f; (void)hipMemcpy(G, h, 16 << 20, hipMemcpyHostToDevice); free(h); }
    (void)hipMalloc(&xc, (size_t)B * 2048 * 4); (void)hipMemset(xc, 0, (size_t)B * 2048 * 4);
    (void)hipMalloc(&q, 2048 * 4); (void)hipMemset(q, 0, 2048 * 4);
    (void)hipMalloc(&idx, (size_t)B * 8);
    { unsigned long long *h = (unsigned long long *)malloc((size_t)B * 8); for (unsigned i = 0; i < B; ++i) { unsigned long long w = 0; for (int m = 0; m < 8; ++m) w |= (unsigned long long)(rand() & 255) << (8 * m); h[i] = w; } (void)hipMemcpy(idx, h, (size_t)B * 8, hipMemcpyHostToDevice); free(h); }
    const unsigned blocks = B * 8 / 4;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&ent, (size_t)B * 8 * 16); (void)hipMalloc(&S, (size_t)B * 8 * 16 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define RUN(name, ...)                                                                                            \
    for (int rep = 0; rep < 3; ++rep) {                                                                           \
        (void)hipEventRecord(e0);                                                                                 \
        k<__VA_ARGS__><<<blocks, 256>>>(G, idx, xc, q, out, ent, S);                                              \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                                  \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                         \
        if (rep == 2) printf("%-28s %.3f ms  %.1f TB/s of row segments\n", name, ms, blocks * 4.0 * 7 * 1024 / ms / 1e9); \
    }
    RUN("pattern alone", false, 0, false, false)
    RUN("+ small output", false, 0, false, true)
    RUN("+ idx (dependent rows)", true, 0, false, true)
    RUN("+ Q", true, 0, true, true)
    RUN("+ x.C from HBM", true, 1, true, true)
    RUN("+ selection (everything)", true, 1, true, true, 1)
    RUN("selection, x.C plain load", true, 2, true, true, 1)
    RUN("sel, x.C plain, prefetch +256", true, 2, true, true, 1, 256)
    RUN("sel, x.C plain, prefetch +512", true, 2, true, true, 1, 512)
    RUN("sel, x.C plain, prefetch +1024", true, 2, true, true, 1, 1024)
    RUN("sel, x.C plain, prefetch +2048", true, 2, true, true, 1, 2048)
    RUN("sel, x.C plain, prefetch +4096", true, 2, true, true, 1, 4096)
    RUN("sel, x.C nt, prefetch +1024", true, 1, true, true, 1, 1024)
    RUN("delay 10, no x.C", true, 0, true, true, 10)
    RUN("delay 10, x.C from HBM", true, 1, true, true, 10)
    RUN("delay 20, no x.C", true, 0, true, true, 20)
    RUN("delay 20, x.C from HBM", true, 1, true, true, 20)
    RUN("delay 40, no x.C", true, 0, true, true, 40)
    RUN("delay 40, x.C from HBM", true, 1, true, true, 40)
    RUN("sel, x.C codebook-major (HBM)", true, 10, true, true, 1)
    RUN("sel, x.C 16 MB window, L2 hits", true, 9, true, true, 1)
    RUN("selection, x.C window 16 MB", true, 3, true, true, 1)
    RUN("selection, x.C window 32 MB", true, 4, true, true, 1)
    RUN("selection, x.C window 64 MB", true, 5, true, true, 1)
    RUN("selection, x.C window 128 MB", true, 6, true, true, 1)
    RUN("selection, x.C window 256 MB", true, 7, true, true, 1)
    RUN("selection, no x.C", true, 0, true, true, 1)
    RUN("pattern + x.C only", false, 1, false, false)
#define RUNL(name, V, A)                                                                                          \
    for (int rep = 0; rep < 3; ++rep) {                                                                           \
        (void)hipEventRecord(e0);                                                                                 \
        kl<V, A><<<blocks / V, 256>>>(G, idx, xc, q, ent, S);                                                     \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                                  \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                         \
        if (rep == 2) printf("%-28s %.3f ms  %.1f TB/s of row segments\n", name, ms, blocks * 4.0 * 7 * 1024 / ms / 1e9); \
    }
#define RUNX(name, M)                                                                                             \
    for (int rep = 0; rep < 3; ++rep) {                                                                           \
        (void)hipEventRecord(e0);                                                                                 \
        kx<M><<<blocks, M == 3 ? 320 : 256>>>(G, idx, xc, q, ent, S);                                             \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                                  \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                         \
        if (rep == 2) printf("%-28s %.3f ms  %.1f TB/s of row segments\n", name, ms, blocks * 4.0 * 7 * 1024 / ms / 1e9); \
    }
    RUNX("x.C first (shipped order)", 0)
    RUNX("x.C after the rows", 1)
    RUNX("x.C by LDS-DMA", 2)
    RUNX("x.C by a fifth wave", 3)
    RUNL("loop V=1, no prefetch", 1, 0)
    RUNL("loop V=2, x.C 1 ahead", 2, 1)
    RUNL("loop V=4, x.C 1 ahead", 4, 1)
    RUNL("loop V=4, x.C 2 ahead", 4, 2)
    RUNL("loop V=8, x.C 1 ahead", 8, 1)
    RUNL("loop V=8, x.C 2 ahead", 8, 2)
    RUNL("loop V=4, no prefetch", 4, 0)
    return 0;
}
