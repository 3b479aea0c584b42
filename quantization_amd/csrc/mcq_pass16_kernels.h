// mcq_pass16_kernels.h -- the refinement passes of 16 (or 8) codebooks of 16 entries (QuantizerTrainer's first phase at 8 (4) bytes per
// frame, /root/reference/quantization/quantization.py:308-547 at K = 16) with the GRAM MATRIX RESIDENT IN LDS.  (Described for N = 16; with
// 8 codebooks there are 28 blocks, two workgroups per CU and one combine level less.)
//
// This is the one shape where "codebooks staged once in LDS" (BASELINE.json north_star) is literally possible for the table form:
// G is 256 x 256, symmetric bit for bit, and a pass never reads a same-codebook entry off the diagonal, so the 120 blocks
// (n, m), n < m, of 16 x 16 floats -- 122,880 bytes -- plus the diagonal (E / R) hold everything a pass reads.  One persistent
// workgroup of eight waves per CU fills them by LDS-DMA; each WAVE then carries one vector at a time through ALL passes of the
// call -- E / R, stage 0, the four combine levels, the winner's walk -- with every Gram read a ds_read, the candidate lists in 4 KB
// of wave-private LDS, the indexes in registers between passes: one launch per call instead of five per pass, no list traffic
// through global memory, no dependent L2 round trips (the separate kernels' waves live 3.6 us at 4,096 vectors, nearly all of it
// waiting for L2-hit gathers and for the lists the launch before wrote).
//
// Same arithmetic as the separate kernels, expression for expression (oracle/mcq_oracle.c, "TABLE FORM"): the sums of stage 0
// in ascending m, the leaf ((g - u) - v) + w, the group tables ((t00 + t01) + t10) + t11 (formed one child table at a time, the
// partial sums in registers in exactly that order), the scores ((Sx + Sy) - E) + 2 T, the shortlists as sets in ascending
// position, E / R of tf_er_wave.  tests/test_gpu_parity.py runs it against the oracle on every fixture of this shape.
#pragma once
#include "mcq_tf_kernels.h"

namespace mcq {

struct Pass16Args {
    const float *G;        // [256][256] Gram matrix of the centered scaled centers
    const float *XC;       // [B][256]   x . C products of the call
    const float *xx;       // [B]        |x - mean|^2
    const float *Q;        // [256]      |C - mean|^2
    uint8_t *idx;          // [B][16]    indexes, refined in place
    long B;
    int iters;
    int64_t *out_i64;      // optional: the result as int64 [B][16]
    uint8_t *out_u8;       // optional: the result as bytes [B][16]
};

constexpr int kP16Waves = 8;
template <int N> constexpr int p16_blocks() { return N * (N - 1) / 2; }                           // codebook pairs n < m
template <int N> constexpr int p16_gram_floats() { return p16_blocks<N>() * 256 + 2 * N * 16; }   // blocks, diagonal, Q
// wave-private scratch (byte offsets)
constexpr int kP16Ent0 = 0;          // u8  [16][8]     level-0 lists: entries
constexpr int kP16Pos1 = 128;        // u8  [8][8][2]   level-1 lists: positions in the halves' lists
constexpr int kP16Pos2 = 256;        // u8  [4][16][2]
constexpr int kP16Pos3 = 384;        // u8  [2][16][2]
constexpr int kP16S0 = 512;          // f32 [16][8]
constexpr int kP16S1 = 1024;         // f32 [8][8]
constexpr int kP16S2 = 1280;         // f32 [4][16]
constexpr int kP16S3 = 1536;         // f32 [2][16]
constexpr int kP16T1 = 1664;         // f32 [8][9]      one level-1 table at a time (rows of 9: odd stride)
constexpr int kP16T2 = 1952;         // f32 [16][17]    one level-2 table at a time
constexpr int kP16Sel = 3040;        // u64 [128]       selection scratch
constexpr int kP16Scratch = 4096;
template <int N> constexpr int p16_lds_bytes() { return p16_gram_floats<N>() * 4 + kP16Waves * kP16Scratch; }      // 157,696 (N = 16), 62,464 (N = 8)
static_assert(kP16Sel + kSelectLdsU64 * 8 <= kP16Scratch && kP16T2 + 16 * 17 * 4 <= kP16Sel && kP16T1 + 8 * 9 * 4 <= kP16T2, "");

__device__ __forceinline__ int shfl_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }

// first float of block (n, m), n < m
template <int N>
__device__ __forceinline__ int p16_blk(int n, int m) { return ((n * (2 * N - 1 - n)) / 2 + m - n - 1) * 256; }

// G[(n, i)][(m, j)], n != m, from the triangular store (G is symmetric bit for bit)
template <int N>
__device__ __forceinline__ float p16_g(const float *Gt, int n, int i, int m, int j) {
    const bool lt = n < m;
    const int lo = lt ? n : m, hi = lt ? m : n, a = lt ? i : j, b = lt ? j : i;
    return Gt[p16_blk<N>(lo, hi) + a * 16 + b];
}

// leaf entry D[n][m] (n < m) for the entries ea of n and eb of m; on / om the current entries: ((g - u) - v) + w
__device__ __forceinline__ float p16_leaf(const float *blk, int ea, int eb, int on, int om) {
    const float g = blk[ea * 16 + eb], u = blk[ea * 16 + om], v = blk[on * 16 + eb], w = blk[on * 16 + om];
    return ((g - u) - v) + w;
}

// T_1[X][Y][i][j] of two level-1 groups X < Y (pairs of codebooks) for this lane's candidates i of X and j of Y
template <int N>
__device__ __forceinline__ float p16_t1(const float *Gt, const uint8_t *ent0, const uint8_t *pos1, int e, int X, int Y, int i, int j) {
    const unsigned pi = *reinterpret_cast<const uint16_t *>(pos1 + (X * 8 + i) * 2), pj = *reinterpret_cast<const uint16_t *>(pos1 + (Y * 8 + j) * 2);
    const int i0 = pi & 0xff, i1 = pi >> 8, j0 = pj & 0xff, j1 = pj >> 8;
    const int ea0 = ent0[(2 * X) * 8 + i0], ea1 = ent0[(2 * X + 1) * 8 + i1], eb0 = ent0[(2 * Y) * 8 + j0], eb1 = ent0[(2 * Y + 1) * 8 + j1];
    const int on0 = __builtin_amdgcn_readlane(e, 2 * X), on1 = __builtin_amdgcn_readlane(e, 2 * X + 1);
    const int om0 = __builtin_amdgcn_readlane(e, 2 * Y), om1 = __builtin_amdgcn_readlane(e, 2 * Y + 1);
    const float d00 = p16_leaf(Gt + p16_blk<N>(2 * X, 2 * Y), ea0, eb0, on0, om0);
    const float d01 = p16_leaf(Gt + p16_blk<N>(2 * X, 2 * Y + 1), ea0, eb1, on0, om1);
    const float d10 = p16_leaf(Gt + p16_blk<N>(2 * X + 1, 2 * Y), ea1, eb0, on1, om0);
    const float d11 = p16_leaf(Gt + p16_blk<N>(2 * X + 1, 2 * Y + 1), ea1, eb1, on1, om1);
    return ((d00 + d01) + d10) + d11;
}

template <int N>
__global__ void __launch_bounds__(64 * kP16Waves)
k_tf_pass16(Pass16Args a) {
    static_assert(N == 16 || N == 8, "");
    constexpr int NK = N * 16, XS = NK / 64, NT = N * N / 64;      // rows of G, x.C slots per lane, Gram terms of E / R per lane
    extern __shared__ __attribute__((aligned(16))) float p16_smem[];
    float *Gt = p16_smem;                        // [120][16][16]
    float *Gd = Gt + p16_blocks<N>() * 256;      // [NK] G[r][r]
    float *Qs = Gd + NK;                         // [NK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *ws = reinterpret_cast<char *>(Qs + NK) + wave * kP16Scratch;
    uint8_t *ent0 = reinterpret_cast<uint8_t *>(ws + kP16Ent0), *pos1 = reinterpret_cast<uint8_t *>(ws + kP16Pos1);
    uint8_t *pos2 = reinterpret_cast<uint8_t *>(ws + kP16Pos2), *pos3 = reinterpret_cast<uint8_t *>(ws + kP16Pos3);
    float *S0 = reinterpret_cast<float *>(ws + kP16S0), *S1 = reinterpret_cast<float *>(ws + kP16S1);
    float *S2 = reinterpret_cast<float *>(ws + kP16S2), *S3 = reinterpret_cast<float *>(ws + kP16S3);
    float *T1a = reinterpret_cast<float *>(ws + kP16T1), *T2a = reinterpret_cast<float *>(ws + kP16T2);
    u64 *sel = reinterpret_cast<u64 *>(ws + kP16Sel);

    // ---- the Gram blocks, global -> LDS without passing through registers: one 1 KB block per DMA (lane l = row l / 4, floats
    // 4 (l % 4) .. + 3 of the block's row), the blocks dealt round the waves
    {
        const unsigned vrow = (unsigned)(lane >> 2), vpart = (unsigned)(lane & 3);
        int cnt = 0;
        for (int n = 0; n < N - 1; ++n)
            for (int m = n + 1; m < N; ++m, ++cnt) {
                if ((cnt & (kP16Waves - 1)) != wave) continue;
                const unsigned go = (((unsigned)(n * 16) + vrow) * (unsigned)NK + (unsigned)(m * 16) + 4u * vpart) * 4u;
                const unsigned d = (unsigned)(size_t)Gt + (unsigned)cnt * 1024u;
                asm volatile("s_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[p]" : : [vo] "v"(go), [p] "s"(a.G), [d] "s"(d) : "memory");
            }
        if (tid < NK) {
            Gd[tid] = a.G[(size_t)tid * NK + tid];
            Qs[tid] = a.Q[tid];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int q4 = lane >> 4, k16 = lane & 15;
    for (long b = (long)blockIdx.x * kP16Waves + wave; b < a.B; b += (long)gridDim.x * kP16Waves) {
        // the vector's inputs: its 256 x.C products (slot r of lane l: row 64 r + l), its indexes (lane n < 16), |x|^2
        float xcv[XS];
#pragma unroll
        for (int r = 0; r < XS; ++r) xcv[r] = a.XC[(size_t)b * NK + 64 * r + lane];
        int e = lane < N ? (int)a.idx[b * N + lane] & 15 : 0;
        const float xxb = a.xx[b];
        for (int pass = 0; pass < a.iters; ++pass) {
            // ---------------------------------------------------------------- E, R (the arithmetic of tf_er_wave)
            float E, Rv;
            {
                float gt[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t = lane + 64 * j, m = t / N, m2 = t % N;     // term t = m * N + m2
                    const int om = shfl_i(e, m), om2 = shfl_i(e, m2);
                    const bool dg = m == m2;
                    const float gd = Gd[m * 16 + om];
                    const float go = p16_g<N>(Gt, m, om, dg ? (m ^ 1) : m2, om2);   // (the diagonal lanes read a valid address and drop it)
                    gt[j] = dg ? gd : go;
                }
                // XC[o_m] in lane m: row m * 16 + o_m of the vector's products = slot m / 4 of lane (m * 16 + o_m) % 64
                const int qrow = (lane < N ? lane : 0) * 16 + e;
                float xt = 0.f;
#pragma unroll
                for (int r = 0; r < XS; ++r) {
                    const float v = shfl_f(xcv[r], qrow & 63);
                    xt = ((qrow >> 6) == r) ? v : xt;
                }
                xt = lane < N ? xt : 0.f;
                float gp = gt[0];
#pragma unroll
                for (int j = 1; j < NT; ++j) gp = gp + gt[j];
                const float gsum = wave_sum_butterfly(gp), xsum = wave_sum_butterfly(xt);
                E = (gsum - 2.0f * xsum) + xxb;
                const int n = lane < N ? lane : 0;
                float col = 0.f, gnn = 0.f;
#pragma unroll
                for (int m = 0; m < N; ++m) {
                    const float v = shfl_f(gt[(m * N) / 64], (m * N + n) & 63);
                    col = (m == 0) ? v : col + v;
                    if (m == n) gnn = v;
                }
                const float xo = col - xt;
                Rv = (E - 2.0f * xo) + gnn;                                  // R[n] in lane n < 16
            }
            // ---------------------------------------------------------------- stage 0: four codebooks per round, lane = (n - 4 r, k)
#pragma unroll
            for (int r = 0; r < N / 4; ++r) {
                const int n = 4 * r + q4;
                float t = 0.f;
                bool started = false;
#pragma unroll
                for (int m = 0; m < N; ++m) {
                    const int om = __builtin_amdgcn_readlane(e, m);
                    const bool use = m != n;
                    const float gv = p16_g<N>(Gt, m, om, use ? n : (n ^ 1), k16);      // row (m, o_m), column (n, k); own codebook: read and dropped
                    const float sum = t + gv;
                    t = use ? (started ? sum : gv) : t;
                    started = started || use;
                }
                const float X = t - xcv[r];
                const float Rn = shfl_f(Rv, n);
                const float sv = (Rn + Qs[64 * r + lane]) + 2.0f * X;
                // the 8 smallest of the row's 16 by (value, entry): rank within the DPP row, listed in ascending entry
                const uint32_t hi = ord32(sv), lo = (uint32_t)k16;
                const u64 key = ((u64)hi << 32) | lo;
                int rnk = 0;
#define MCQ_ROR(rr)                                                                                                     \
    {                                                                                                                   \
        const u64 o = ((u64)(uint32_t)dpp_i<0x120 + rr>((int)hi) << 32) | (uint32_t)dpp_i<0x120 + rr>((int)lo);         \
        rnk += (o < key) ? 1 : 0;                                                                                       \
    }
                MCQ_ROR(1) MCQ_ROR(2) MCQ_ROR(3) MCQ_ROR(4) MCQ_ROR(5) MCQ_ROR(6) MCQ_ROR(7) MCQ_ROR(8)
                MCQ_ROR(9) MCQ_ROR(10) MCQ_ROR(11) MCQ_ROR(12) MCQ_ROR(13) MCQ_ROR(14) MCQ_ROR(15)
#undef MCQ_ROR
                const bool take = rnk < 8;
                const u64 tm = __ballot(take);
                const uint32_t rowbits = (uint32_t)(tm >> (lane & 48)) & 0xffffu;
                const int dst = __popc(rowbits & ((1u << k16) - 1u));
                if (take) {
                    ent0[n * 8 + dst] = (uint8_t)k16;
                    S0[n * 8 + dst] = sv;
                }
            }
            wave_lds_fence();
            // ---------------------------------------------------------------- level 0: the eight sibling pairs of codebooks
            const int i8 = lane >> 3, j8 = lane & 7;
            for (int g = 0; g < N / 2; ++g) {
                const int n = 2 * g, m = n + 1;
                const int on = __builtin_amdgcn_readlane(e, n), om = __builtin_amdgcn_readlane(e, m);
                const float d = p16_leaf(Gt + p16_blk<N>(n, m), ent0[n * 8 + i8], ent0[m * 8 + j8], on, om);
                float sv[1] = {((S0[n * 8 + i8] + S0[m * 8 + j8]) - E) + 2.0f * d};
                int sp[1] = {lane};
                bool has;
                int dst, op;
                float ov;
                wave_select_set<1>(sv, sp, 8, 64, sel, has, dst, ov, op);
                if (has) {
                    pos1[(g * 8 + dst) * 2] = (uint8_t)(op >> 3);
                    pos1[(g * 8 + dst) * 2 + 1] = (uint8_t)(op & 7);
                    S1[g * 8 + dst] = ov;
                }
            }
            wave_lds_fence();
            // ---------------------------------------------------------------- level 1: the four sibling pairs of level-1 groups
            for (int h = 0; h < N / 4; ++h) {
                const int X = 2 * h, Y = X + 1;
                const float t = p16_t1<N>(Gt, ent0, pos1, e, X, Y, i8, j8);
                float sv[1] = {((S1[X * 8 + i8] + S1[Y * 8 + j8]) - E) + 2.0f * t};
                int sp[1] = {lane};
                bool has;
                int dst, op;
                float ov;
                wave_select_set<1>(sv, sp, 16, 64, sel, has, dst, ov, op);
                if (has) {
                    pos2[(h * 16 + dst) * 2] = (uint8_t)(op >> 3);
                    pos2[(h * 16 + dst) * 2 + 1] = (uint8_t)(op & 7);
                    S2[h * 16 + dst] = ov;
                }
            }
            wave_lds_fence();
            // ---------------------------------------------------------------- level 2: the two sibling pairs of level-2 groups
            // candidates p = 4 lane + v: i = lane / 4 of the left list, j = 4 (lane % 4) + v of the right one
            const int i16 = lane >> 2, jb = 4 * (lane & 3);
            int win = 0;
            for (int q = 0; q < N / 8; ++q) {
                const int P = 2 * q, Q2 = P + 1;                  // level-2 groups; their halves are the level-1 groups 2P, 2P+1 / 2Q2, 2Q2+1
                const unsigned pi = *reinterpret_cast<const uint16_t *>(pos2 + (P * 16 + i16) * 2);
                const uint64_t pj = *reinterpret_cast<const uint64_t *>(pos2 + (Q2 * 16 + jb) * 2);      // four (j0, j1) pairs
                float part[4];
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    const int aa = tb >> 1, cc = tb & 1;
                    T1a[i8 * 9 + j8] = p16_t1<N>(Gt, ent0, pos1, e, 2 * P + aa, 2 * Q2 + cc, i8, j8);
                    wave_lds_fence();
                    const int ri = (int)((pi >> (8 * aa)) & 0xffu);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int rj = (int)((pj >> (16 * v + 8 * cc)) & 0xffull);
                        const float val = T1a[ri * 9 + rj];
                        part[v] = (tb == 0) ? val : part[v] + val;
                    }
                    wave_lds_fence();
                }
                float sv[4];
                int sp[4];
                const float sx = S2[P * 16 + i16];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    sv[v] = ((sx + S2[Q2 * 16 + jb + v]) - E) + 2.0f * part[v];
                    sp[v] = 4 * lane + v;
                }
                if constexpr (N == 8) {
                    // eight codebooks: these 256 are the whole vector's candidates -> the winner
                    float bv = INFINITY;
                    int bp = kBigPos;
#pragma unroll
                    for (int v = 0; v < 4; ++v) lexmin(bv, bp, sv[v], sp[v]);
                    wave_lexmin(bv, bp);
                    win = __builtin_amdgcn_readfirstlane(bp);
                    if (win > 255) win = 255;                      // only reachable with NaN keys
                } else {
                    bool has;
                    int dst, op;
                    float ov;
                    wave_select_set<4>(sv, sp, 16, 256, sel, has, dst, ov, op);
                    if (has) {
                        pos3[(q * 16 + dst) * 2] = (uint8_t)(op >> 4);
                        pos3[(q * 16 + dst) * 2 + 1] = (uint8_t)(op & 15);
                        S3[q * 16 + dst] = ov;
                    }
                }
            }
            wave_lds_fence();
            // ---------------------------------------------------------------- level 3: the two groups of eight codebooks -> the winner
            if constexpr (N == 16) {
                const unsigned pi3 = *reinterpret_cast<const uint16_t *>(pos3 + i16 * 2);
                const uint64_t pj3 = *reinterpret_cast<const uint64_t *>(pos3 + (16 + jb) * 2);
                float part[4];
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    const int xc = t2 >> 1, yc = t2 & 1;          // level-2 table T_2[xc][2 + yc]
                    const unsigned pa = *reinterpret_cast<const uint16_t *>(pos2 + (xc * 16 + i16) * 2);
                    const uint64_t pb = *reinterpret_cast<const uint64_t *>(pos2 + ((2 + yc) * 16 + jb) * 2);
                    float p2[4];
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb) {
                        const int aa = tb >> 1, cc = tb & 1;
                        T1a[i8 * 9 + j8] = p16_t1<N>(Gt, ent0, pos1, e, 2 * xc + aa, 4 + 2 * yc + cc, i8, j8);
                        wave_lds_fence();
                        const int ri = (int)((pa >> (8 * aa)) & 0xffu);
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int rj = (int)((pb >> (16 * v + 8 * cc)) & 0xffull);
                            const float val = T1a[ri * 9 + rj];
                            p2[v] = (tb == 0) ? val : p2[v] + val;
                        }
                        wave_lds_fence();
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) T2a[i16 * 17 + jb + v] = p2[v];
                    wave_lds_fence();
                    const int ri = (int)((pi3 >> (8 * xc)) & 0xffu);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int rj = (int)((pj3 >> (16 * v + 8 * yc)) & 0xffull);
                        const float val = T2a[ri * 17 + rj];
                        part[v] = (t2 == 0) ? val : part[v] + val;
                    }
                    wave_lds_fence();
                }
                float bv = INFINITY;
                int bp = kBigPos;
                const float sx = S3[i16];
#pragma unroll
                for (int v = 0; v < 4; ++v) lexmin(bv, bp, ((sx + S3[16 + jb + v]) - E) + 2.0f * part[v], 4 * lane + v);
                wave_lexmin(bv, bp);
                win = __builtin_amdgcn_readfirstlane(bp);
                if (win > 255) win = 255;                          // only reachable with NaN keys
            }
            // ---------------------------------------------------------------- the winner's leaves (tf_emit): lane n walks down the tree
            {
                int en = 0;
                if (lane < N) {
                    const int n = lane;
                    int p = (n >> (N == 16 ? 3 : 2)) ? (win & 15) : (win >> 4);               // position in the top list of the vector's half
                    if constexpr (N == 16) p = pos3[((n >> 3) * 16 + p) * 2 + ((n >> 2) & 1)];   // -> level-2 list of group n / 4
                    p = pos2[((n >> 2) * 16 + p) * 2 + ((n >> 1) & 1)];                       // -> level-1 list of group n / 2
                    p = pos1[((n >> 1) * 8 + p) * 2 + (n & 1)];                               // -> level-0 list of codebook n
                    en = ent0[n * 8 + p];
                }
                e = en;
            }
            wave_lds_fence();
        }
        if (lane < N) {
            a.idx[b * N + lane] = (uint8_t)e;
            if (a.out_i64) a.out_i64[b * N + lane] = e;
            if (a.out_u8) a.out_u8[b * N + lane] = (uint8_t)e;
        }
    }
}

}  // namespace mcq
