// mcq_tf_kernels.h -- gfx950 kernels of the TABLE FORM of the refinement pass (2 <= N <= 16 codebooks).
//
// Every inner product of _refine_indexes (/root/reference/quantization/quantization.py:403-416, :533-535) is linear
// in codebook rows, so it is READ from two tables instead of being recomputed per vector and pass:
//   G[r][c]  = dot16(C[r], C[c])   Gram matrix of the N*K scaled centers, part of the prepared state
//   XC[b][r] = dot16(C[r], x[b])   one GEMM per encode call (k_gemm8s<MODE_XC>)
// Numeric contract: oracle/mcq_oracle.c, "TABLE FORM" (single IEEE fp32 adds / subs in the order written there).
//   stage 0   X = (sum over m != n ascending of G[(m, idx_m)][(n, k)]) - XC[b][(n, k)];  S = (R + Q) + 2 X
//   leaf      D[n][m][i][j] = ((G[s_n,i][s_m,j] - G[s_n,i][o_m]) - G[o_n][s_m,j]) + G[o_n][o_m]
//   group     T_l[X][Y][i][j] = ((T_h[X0][Y0][i0][j0] + T_h[X0][Y1][i0][j1]) + T_h[X1][Y0][i1][j0]) + T_h[X1][Y1][i1][j1]
//   combine   S' = ((S_X[i] + S_Y[j]) - E) + 2 T_l[X][Y][i][j]        (siblings X = 2g, Y = 2g + 1)
// Lists: level v holds, per group of 2^v codebooks, kc[v] = K_cutoff(K, 2^v) candidates; level 0 as codebook
// entries `ent`, level v >= 1 as the pair of positions in the two halves' lists `pos`.  The result is read off
// the position tree (tf_emit).  What moves per vector and pass is a few thousand 4-byte table reads served by
// the XCD's L2 (workgroup id mod #groups picks the group / table, so an XCD only touches its own blocks of G)
// instead of 768 KB of gathered codebook rows: no MFMA, no operand staging.
#pragma once
#include "mcq_kernels.h"

#include <type_traits>

namespace mcq {

constexpr int kTfLevels = 6;   // lists of candidates over 1, 2, 4, 8, 16, 32 codebooks (N <= 64)

// A codebook entry is held in one byte up to 256 entries per codebook (what QuantizerTrainer produces and what encode() packs,
// quantization.py:266-271) and in two bytes above (Quantizer(codebook_size = 512 / 1024), as_bytes = False); the kernels that
// touch entries take the type as `CT`.  The one-byte instantiations are the tuned ones; the two-byte ones take the plain paths.
template <int K>
using tf_code_of = std::conditional_t<(K > 256), uint16_t, uint8_t>;

struct TfLists {
    uint8_t *ent;               // [B][N][kc[0]]            level-0 lists: codebook entries (CT: one or two bytes each)
    uint8_t *pos[kTfLevels];    // [B][N >> v][kc[v]][2]    level v >= 1: positions in the halves' lists
    float *S[kTfLevels];        // [B][N >> v][kc[v]]       scores
    int kc[kTfLevels];
    // set for the LAST pass of a call when the result can leave in its final form: the winner's entries also go straight
    // to the caller's arrays (int64 [B][N] and / or uint8 [B][N]) and the encode tail needs no launch of its own
    int64_t *out_i64;
    uint8_t *out_u8;
    // set for every pass but the last: the wave that emits a vector's new indexes also forms E and R[n] of the NEXT pass from
    // them (tf_er_wave: the arithmetic of k_tf_er), so that pass needs neither k_tf_gram_terms nor k_tf_er
    const float *erG, *erXC, *erxx;
    float *erE, *erR;
    int erK;
};

__device__ __forceinline__ float shfl_f(float v, int src) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}

// Phase stamps of sampled waves (a DEBUG build only: hipcc -DMCQ_STAMPS, tools/exp_stamps.py): s_memtime at the phase boundaries
// of the three large pass kernels, every 64th workgroup, read back through mcq_debug_stamps.  Compiled out otherwise (an empty
// struct whose calls vanish); it never changes a result.
#ifdef MCQ_STAMPS
constexpr int kStampSlots = 8, kStampWaves = 1 << 14, kStampKernels = 4;
__device__ unsigned long long g_stamps[kStampKernels * kStampWaves * kStampSlots];
struct Stamps {
    unsigned long long t[kStampSlots];
    __device__ __forceinline__ Stamps() { for (int i = 0; i < kStampSlots; ++i) t[i] = 0; t[0] = __builtin_amdgcn_s_memtime(); }
    // everything requested so far has arrived when the stamp is taken (the build exists to see where a wave's time goes)
    __device__ __forceinline__ void at(int k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t[k] = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void flush(int kernel, unsigned wg, int every) {
        if (wg % (unsigned)every != 0 || (threadIdx.x & 63) != 0 || (threadIdx.x >> 6) != 0) return;
        const unsigned slot = (wg / (unsigned)every) & (kStampWaves - 1);
        unsigned long long *d = g_stamps + ((size_t)kernel * kStampWaves + slot) * kStampSlots;
        t[kStampSlots - 1] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < kStampSlots; ++i) d[i] = t[i];
    }
};
#else
struct Stamps {
    __device__ __forceinline__ void at(int) {}
    __device__ __forceinline__ void flush(int, unsigned, int) {}
};
#endif

// (|x|^2, the constant of E, is formed by k_fix_rows while it turns the frames into limb planes)

// --------------------------------------------------------------------- E, R
// The N*N Gram entries G[o_m][o_m2] of a vector lie all over the matrix: gathered by one wave per vector they miss the
// 4 MB L2s (16 MB matrix at 8 x 256: 203 MB fetched from the fabric per launch for 21 MB of useful reads, 0.061 ms).
// k_tf_gram_terms gathers them XCD by XCD instead: workgroup id mod N = m, a wave takes 64 / N vectors x the N entries of
// row block m, so an XCD only ever touches the rows of its own codebooks (2 MB); the terms go to gterms[b][m][m2] and
// k_tf_er reads its 64 terms as one coalesced line.
template <int N, typename CT = uint8_t>
__global__ void __launch_bounds__(256)
k_tf_gram_terms(const float *__restrict__ G, const CT *__restrict__ idx, long B, int K, float *__restrict__ gterms,
                const int *__restrict__ nact) {
    constexpr int VW = 64 / N;                     // vectors per wave
    if (nact) B = *nact;
    const int m = (int)(blockIdx.x % N);
    const long b = ((long)(blockIdx.x / N) * 4 + (threadIdx.x >> 6)) * VW + lane_id() / N;
    const int m2 = lane_id() % N;
    if (b >= B) return;
    const CT *id = idx + b * N;
    const int NK = N * K;
    gterms[(b * N + m) * N + m2] = G[((size_t)(m * K + id[m]) << __builtin_ctz((unsigned)NK)) + m2 * K + id[m2]];
}

// One wave per vector: E = |x_err|^2 and R[n] = |x_err - old_n|^2 (:401-409) from the N*N Gram terms, N entries of XC
// and |x|^2 (oracle "TABLE FORM", E, R).
// The indexes come from memory (`id`) or, when the wave has just chosen them, from lane m of `e_reg` (FROM_REG).
template <int N, bool FROM_REG, typename CT = uint8_t>
__device__ __forceinline__ void tf_er_wave(long b, long bx, const CT *__restrict__ id, int e_reg,
                                           const float *__restrict__ gterms, const float *__restrict__ Gdirect,
                                           const float *__restrict__ XC, const float *__restrict__ xx, int K,
                                           float *__restrict__ E_out, float *__restrict__ R_out) {
    constexpr int NT = (N * N + 63) / 64;          // Gram terms per lane
    const int lane = lane_id();
    const int NK = N * K;
#define MCQ_ER_CODE(m) (FROM_REG ? __builtin_amdgcn_ds_bpermute((m) << 2, e_reg) : (int)id[FROM_REG ? 0 : (m)])
    // term t = m * N + m2 (lane t % 64, slot t / 64) is G[o_m][o_m2]
    float gt[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = lane + 64 * j;
        const int tc = t < N * N ? t : 0;
        // without the XCD-by-XCD launch in front (small batches; the wave that has just chosen the indexes) the Gram entries
        // are gathered here
        const float g = Gdirect ? Gdirect[(size_t)((tc / N) * K + MCQ_ER_CODE(tc / N)) * NK + (tc % N) * K + MCQ_ER_CODE(tc % N)]
                                : gterms[b * (N * N) + tc];
        gt[j] = t < N * N ? g : 0.f;
    }
    const int lm = lane < N ? lane : 0;
    const int clm = MCQ_ER_CODE(lm);
#undef MCQ_ER_CODE
    const float xt = lane < N ? XC[(size_t)bx * NK + lm * K + clm] : 0.f;        // XC[o_m] in lane m
    const float xxb = xx[bx];
    float gp = gt[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) gp = gp + gt[j];
    const float gsum = wave_sum_butterfly(gp), xsum = wave_sum_butterfly(xt);
    const float E = (gsum - 2.0f * xsum) + xxb;
    // lane n < N: R[n] from column n of the N x N block, m ascending
    const int n = lane < N ? lane : 0;
    float col = 0.f, gnn = 0.f;
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const float v = shfl_f(gt[(m * N) / 64], (m * N + n) & 63);   // term m * N + n: its slot is a compile-time function of m
        col = (m == 0) ? v : col + v;
        if (m == n) gnn = v;
    }
    const float xo = col - xt;
    const float Rv = (E - 2.0f * xo) + gnn;
    if (lane < N) R_out[b * N + lane] = Rv;
    if (lane == 0) E_out[b] = E;
}

template <int N, typename CT = uint8_t>
__global__ void __launch_bounds__(256)
k_tf_er(const float *__restrict__ gterms, const float *__restrict__ XC, const CT *__restrict__ idx,
        const float *__restrict__ xx, long B, int K, float *__restrict__ E_out, float *__restrict__ R_out,
        const int *__restrict__ nact, const int *__restrict__ map, const float *__restrict__ Gdirect) {
    if (nact) B = *nact;
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const long bx = map ? (long)map[b] : b;          // row of the per-call arrays (XC, xx)
    tf_er_wave<N, false, CT>(b, bx, idx + b * N, 0, gterms, Gdirect, XC, xx, K, E_out, R_out);
}

// ------------------------------------------------------------------ stage 0
// One wave per (vector, codebook n): the K scores of :418 from N - 1 row segments of G, the vector's XC segment and
// Q, then the first sort-and-truncate (:470-503).  Workgroup id mod N = n: an XCD's L2 holds G[:, segment n] only.
template <int K, int N>
__global__ void __launch_bounds__(256)
k_tf_stage0(const float *__restrict__ G, const float *__restrict__ XC, const tf_code_of<K> *__restrict__ idx,
            const float *__restrict__ R, const float *__restrict__ Q, long B, int keep,
            tf_code_of<K> *__restrict__ ent_out, float *__restrict__ S_out, tf_code_of<K> *__restrict__ idx_final /* N == 1 */,
            const int *__restrict__ nact, const int *__restrict__ map) {
    using CT = tf_code_of<K>;
    constexpr int CB = (int)sizeof(CT);
    constexpr int VPL = (K >= 64) ? K / 64 : 1;
    constexpr int NK = N * K;
    constexpr int CH = (N - 1 < 8) ? (N > 1 ? N - 1 : 1) : 8;    // row segments in flight
    __shared__ u64 sel[4][kSelectLdsU64];
    Stamps stp;
    if (nact) B = *nact;
    // workgroup -> (vector quad, codebook): id mod 8 = the XCD.  Up to 8 codebooks: n = id mod N.  More: the launch runs
    // in N / 8 phases, phase ph covering the codebooks 8 ph .. 8 ph + 7 for every vector, so that an XCD works on ONE column
    // segment of G at a time (4 MB at 16 x 256: with n = id mod 16 its L2 was asked to hold two)
    int n;
    long b;
    if constexpr (N > 8) {
        const unsigned per_phase = gridDim.x / (N / 8);
        const unsigned ph = blockIdx.x / per_phase, r = blockIdx.x - ph * per_phase;
        n = (int)(r & 7u) + 8 * (int)ph;
        b = (long)(r >> 3) * 4 + (threadIdx.x >> 6);
    } else {
        n = blockIdx.x & (N - 1);
        b = (long)(blockIdx.x / N) * 4 + (threadIdx.x >> 6);
    }
    if (b >= B) return;
    const int lane = lane_id();
    const bool act = VPL * lane < K;
    const int k0 = act ? VPL * lane : 0;
    const CT *id = idx + b * N;
    // the vector's own inputs (its x.C segment comes from HBM: the longest latency of the kernel) are requested first,
    // beside the index bytes, not after the Gram rows that depend on those bytes
    const float *xc = XC + ((size_t)(map ? (long)map[b] : b) * NK + n * K + k0);
    const float *q = Q + n * K + k0;
    float xcv[VPL], qv[VPL];
    if constexpr (VPL == 4) {
        const f32x4 x4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xc)), q4 = *reinterpret_cast<const f32x4 *>(q);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xcv[i] = x4[i]; qv[i] = q4[i]; }
    } else if constexpr (VPL % 4 == 0) {      // codebooks of 512 / 1,024 entries
#pragma unroll
        for (int i4 = 0; i4 < VPL / 4; ++i4) {
            const f32x4 x4 = reinterpret_cast<const f32x4 *>(xc)[i4], q4 = reinterpret_cast<const f32x4 *>(q)[i4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xcv[4 * i4 + i] = x4[i]; qv[4 * i4 + i] = q4[i]; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) { xcv[i] = xc[i]; qv[i] = q[i]; }
    }
    // (b is the same for the whole wave, which the compiler cannot see: through readfirstlane this one float is a scalar load
    // instead of a vector load every lane takes part in)
    const float Rv = R[(long)__builtin_amdgcn_readfirstlane((int)b) * N + n];
    __builtin_amdgcn_sched_barrier(0);                      // (the scheduler otherwise sinks the x.C read below the Gram reads)
    // the vector's N index bytes: one scalar load per 8 of them (the vector is the same for the whole wave) instead of N - 1
    // byte loads per lane; the Gram row addresses are then scalar too
    unsigned long long iw[N * CB >= 8 ? N * CB / 8 : 1];
    if constexpr (N * CB >= 8) {
        const unsigned long long *ip =
            reinterpret_cast<const unsigned long long *>(idx) + (size_t)__builtin_amdgcn_readfirstlane((int)b) * (N * CB / 8);
#pragma unroll
        for (int q8 = 0; q8 < N * CB / 8; ++q8) iw[q8] = ip[q8];
    }
    auto code = [&](int m) -> int {
        if constexpr (N * CB >= 8) return (int)((iw[(m * CB) >> 3] >> (8 * ((m * CB) & 7))) & (CB == 1 ? 0xffull : 0xffffull));
        else return (int)id[m];
    };
    float t[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) t[i] = 0.f;               // N == 1: no other codebook, X = 0 - XC
#pragma unroll
    for (int j0 = 0; j0 < N - 1; j0 += CH) {
        float gv[CH][VPL];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = j0 + u < N - 1 ? j0 + u : N - 2;
            const int m = j < n ? j : j + 1;                    // m ascending over the codebooks other than n
            const float *p = G + ((size_t)(m * K + code(m)) * NK + n * K + k0);
            if constexpr (VPL % 4 == 0) {
#pragma unroll
                for (int i4 = 0; i4 < VPL / 4; ++i4) {
                    const f32x4 t4 = reinterpret_cast<const f32x4 *>(p)[i4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[u][4 * i4 + i] = t4[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < VPL; ++i) gv[u][i] = p[i];
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (j0 + u < N - 1) {
#pragma unroll
                for (int i = 0; i < VPL; ++i) t[i] = (j0 + u == 0) ? gv[u][i] : t[i] + gv[u][i];
            }
    }
    float sv[VPL];
    int sp[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float X = t[i] - xcv[i];
        sv[i] = act ? (Rv + qv[i]) + 2.0f * X : INFINITY;
        sp[i] = act ? k0 + i : kBigPos;
    }
    stp.at(1);                                  // all rows in, scores formed
    float ov;
    int op, dst;
    bool has;
    wave_select_set<VPL>(sv, sp, keep, K, sel[threadIdx.x >> 6], has, dst, ov, op);
    stp.at(2);                                  // selection done
    stp.flush(0, blockIdx.x, 16);
    if (N == 1) {                                             // the best entry is the result (:468-469)
        if (lane == 0) idx_final[b] = (CT)op;
        return;
    }
    if (has) {                                                // the list in ascending entry: every survivor from the lane that holds it
        ent_out[(b * N + n) * keep + dst] = (CT)op;
        S_out[(b * N + n) * keep + dst] = ov;
    }
}

// Stage 0 for 16-entry codebooks (the trainer's first phase, K = 16 inference): k_tf_stage0 gives a wave to one (vector,
// codebook) and 16 of its 64 lanes a score; here a wave takes FOUR codebooks n0 .. n0 + 3 of a vector, lane = 16 * (n - n0) + k.
// A row of G then serves all four (256 contiguous bytes), the own-codebook row is read and skipped (m ascending over m != n,
// as in k_tf_stage0: same sums), and the sort-and-truncate of :470-503 is a rank within the lane's DPP row of 16: fifteen
// row rotations of the 64-bit key (score bits || position), no scalar loop, no LDS.  Same results, bit for bit.
template <int N>
__global__ void __launch_bounds__(256)
k_tf_stage0_k16(const float *__restrict__ G, const float *__restrict__ XC, const uint8_t *__restrict__ idx,
                const float *__restrict__ R, const float *__restrict__ Q, long B, int keep,
                uint8_t *__restrict__ ent_out, float *__restrict__ S_out, const int *__restrict__ nact,
                const int *__restrict__ map) {
    static_assert(N >= 4, "");
    constexpr int K = 16, NK = N * K, NG = N / 4;
    if (nact) B = *nact;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long b = w / NG;
    const int n0 = 4 * (int)(w % NG);
    if (b >= B) return;
    const int lane = lane_id();
    const int n = n0 + (lane >> 4), k = lane & 15;
    const float xcv = XC[(size_t)(map ? (long)map[b] : b) * NK + n0 * K + lane];
    const float qv = Q[n0 * K + lane];
    const float Rv = R[b * N + n];
    // the vector's N index bytes through scalar loads (the vector is the same for the whole wave)
    const long bu = __builtin_amdgcn_readfirstlane((int)b);
    unsigned long long iw[N >= 8 ? N / 8 : 1];
    if constexpr (N >= 8) {
        const unsigned long long *ip = reinterpret_cast<const unsigned long long *>(idx) + (size_t)bu * (N / 8);
#pragma unroll
        for (int q8 = 0; q8 < N / 8; ++q8) iw[q8] = ip[q8];
    } else {
        iw[0] = (unsigned long long)reinterpret_cast<const uint32_t *>(idx)[bu];      // N == 4
    }
    auto code = [&](int m) -> int { return (int)((iw[m >> 3] >> (8 * (m & 7))) & 0xffull) & (K - 1); };
    constexpr int CH = N < 8 ? N : 8;                 // rows in flight
    float t = 0.f;
    bool started = false;
#pragma unroll
    for (int m0 = 0; m0 < N; m0 += CH) {
        float gv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) gv[u] = G[(size_t)((m0 + u) * K + code(m0 + u)) * NK + n0 * K + lane];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const bool use = (m0 + u) != n;
            const float sum = t + gv[u];
            t = use ? (started ? sum : gv[u]) : t;
            started = started || use;
        }
    }
    const float X = t - xcv;
    const float sv = (Rv + qv) + 2.0f * X;
    // rank of (sv, k) among the 16 keys of this lane's row
    const uint32_t hi = ord32(sv), lo = (uint32_t)k;
    const u64 key = ((u64)hi << 32) | lo;
    int rnk = 0;
#define MCQ_ROR(r)                                                                                                      \
    {                                                                                                                   \
        const u64 o = ((u64)(uint32_t)dpp_i<0x120 + r>((int)hi) << 32) | (uint32_t)dpp_i<0x120 + r>((int)lo);           \
        rnk += (o < key) ? 1 : 0;                                                                                       \
    }
    MCQ_ROR(1) MCQ_ROR(2) MCQ_ROR(3) MCQ_ROR(4) MCQ_ROR(5) MCQ_ROR(6) MCQ_ROR(7) MCQ_ROR(8)
    MCQ_ROR(9) MCQ_ROR(10) MCQ_ROR(11) MCQ_ROR(12) MCQ_ROR(13) MCQ_ROR(14) MCQ_ROR(15)
#undef MCQ_ROR
    // the survivors (rank < keep) listed in ascending entry: the index is a prefix count within the lane's row of 16
    const bool take = rnk < keep;
    const u64 tm = __ballot(take);
    const uint32_t rowbits = (uint32_t)(tm >> (lane & 48)) & 0xffffu;
    const int dst = __popc(rowbits & ((1u << k) - 1u));
    if (take) {
        ent_out[(b * N + n) * keep + dst] = (uint8_t)k;
        S_out[(b * N + n) * keep + dst] = sv;
    }
}

// ---------------------------------------------------------------- leaf table
// D[n][m] (codebooks n < m) over the two level-0 lists of KC entries: lane holds positions p = VPL*lane + v
// (row i = p / KC, column j = p % KC).  KC*KC core reads plus one read per lane for the 2*KC + 1 border values
// G[s_n,i][o_m] (lanes 0..KC-1), G[o_n][s_m,j] (KC..2KC-1), G[o_n][o_m] (lane 2KC), handed round by ds_bpermute.
template <int KC, typename CT = uint8_t>
__device__ __forceinline__ void tf_leaf(const float *__restrict__ G, int NK, int K, int n, int m,
                                        const CT *__restrict__ en, const CT *__restrict__ em, int old_n,
                                        int old_m, float (&d)[KC * KC / 64]) {
    constexpr int VPL = KC * KC / 64;
    const int lane = lane_id();
    const int i = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
    const uint32_t rown = (uint32_t)(n * K), colm = (uint32_t)(m * K);
    const int nksh = __builtin_ctz((unsigned)NK);      // N and K are powers of two: a row offset is a shift (v_mul_lo_u32 runs at a quarter of the rate)
    // every list byte this lane needs (its row entry, its VPL column entries, its border entry) is requested before the first
    // use: left to the compiler the border bytes were loaded one after the other BEHIND the first wait, and the Gram reads
    // started four memory round trips into the kernel instead of two
    const int bl = lane < 2 * KC ? lane : 2 * KC;
    const CT *bp = bl < KC ? en + bl : em + (bl < 2 * KC ? bl - KC : 0);
    int e_i = en[i], e_b = *bp;
    uint32_t w4 = 0;
    int ej[VPL];
    constexpr bool kPacked = (VPL == 4 && sizeof(CT) == 1);      // the lane's four column entries as one 32-bit word
    if constexpr (kPacked) {
        w4 = *reinterpret_cast<const uint32_t *>(em + j0);
        asm volatile("" : "+v"(e_i), "+v"(e_b), "+v"(w4));
    } else {
#pragma unroll
        for (int v = 0; v < VPL; ++v) ej[v] = em[j0 + v];
        asm volatile("" : "+v"(e_i), "+v"(e_b), "+v"(ej[0]));
    }
    const uint32_t si = rown + (uint32_t)e_i;
    const uint32_t br = bl < KC ? rown + (uint32_t)e_b : rown + (uint32_t)old_n;
    const uint32_t bc = (bl >= KC && bl < 2 * KC) ? colm + (uint32_t)e_b : colm + (uint32_t)old_m;
    float g[VPL];
    if constexpr (kPacked) {
#pragma unroll
        for (int v = 0; v < 4; ++v) g[v] = G[(si << nksh) + colm + ((w4 >> (8 * v)) & 0xffu)];
    } else {
#pragma unroll
        for (int v = 0; v < VPL; ++v) g[v] = G[(si << nksh) + colm + (uint32_t)ej[v]];
    }
    // (the border entries G[s_n,i][o_m] of lanes 0 .. KC - 1 lie in the rows the core reads above have just asked for -- mostly the
    // same lines; read through the symmetry of G as G[o_m][s_n,i], as tf_table1 does for its compact tables, they cost 7 % more
    // L2 requests here at the same time: profiles/r04_pmc_counters.txt of the two builds)
    const float bv = G[(br << nksh) + bc];
    const float u = shfl_f(bv, i), w = shfl_f(bv, 2 * KC);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const float vj = shfl_f(bv, KC + j0 + v);
        d[v] = ((g[v] - u) - vj) + w;
    }
}

// E and R[n] of the next pass for the vector whose new indexes sit in lanes 0 .. N - 1 of `e` (tf_emit).  A function of its
// own (not inlined: the three instantiations inside every last-combine kernel sent the compiler's CFG simplification into
// a crash).
__device__ __attribute__((noinline)) void tf_er_next(const float *G, const float *XC, const float *xx, int K, float *E, float *R,
                                                     long b, int N, int e) {
    if (N == 8) tf_er_wave<8, true, uint8_t>(b, b, nullptr, e, nullptr, G, XC, xx, K, E, R);
    else if (N == 4) tf_er_wave<4, true, uint8_t>(b, b, nullptr, e, nullptr, G, XC, xx, K, E, R);
    else if (N == 16) tf_er_wave<16, true, uint8_t>(b, b, nullptr, e, nullptr, G, XC, xx, K, E, R);
}

// The winner's leaves, codebook by codebook (:468-469): lane n walks down the position tree.
// `win` = position a*kc + b of the winner among the pairs of the two top-level lists.
template <typename CT>
__device__ __forceinline__ void tf_emit(const TfLists &L, long b, int N, int nlev, int win, CT *__restrict__ idx_out) {
    const int n = lane_id();
    int e = 0;
    if (n < N) {
        int v = nlev - 1;
        int g = n >> v;
        int p = (g & 1) ? win % L.kc[v] : win / L.kc[v];
        while (v > 0) {
            const int child = (n >> (v - 1)) & 1;
            p = L.pos[v][((b * (N >> v) + g) * L.kc[v] + p) * 2 + child];
            --v;
            g = n >> v;
        }
        e = reinterpret_cast<const CT *>(L.ent)[(b * N + n) * L.kc[0] + p];
        idx_out[b * N + n] = (CT)e;
        if (L.out_i64) L.out_i64[b * N + n] = e;
        if (L.out_u8) L.out_u8[b * N + n] = (uint8_t)e;      // (one-byte entries only: the host leaves it null otherwise)
    }
    if (L.erE) tf_er_next(L.erG, L.erXC, L.erxx, L.erK, L.erE, L.erR, b, N, e);      // E, R[n] of the next pass, from the indexes in lanes 0 .. N - 1
}

// select `keep` of the wave's scores and write the next level's list (or, for the last combine, the result)
template <int VPL, typename CT = uint8_t, int LAYOUT = kLaneMajor>
__device__ __forceinline__ void tf_finish(const float (&sv)[VPL], const int (&sp)[VPL], int keep, int KCin, u64 *scratch,
                                          const TfLists &L, int vout /* level of the list written */, long b, int N,
                                          int gout, CT *__restrict__ idx_final) {
    if (idx_final != nullptr) {     // one group left, keep == 1: the smallest (score, position) is the result
        // (wave_select's rule for one winner -- smallest (score, position), NaN never taken -- without its general
        // bookkeeping of the previous winner: ten instructions per candidate there)
        float bv = INFINITY;
        int bp = kBigPos;
#pragma unroll
        for (int i = 0; i < VPL; ++i) lexmin(bv, bp, sv[i], sp[i]);
        wave_lexmin(bv, bp);
        int win = __builtin_amdgcn_readfirstlane(bp);
        if (win > KCin * KCin - 1) win = KCin * KCin - 1;          // only reachable with NaN keys
        tf_emit(L, b, N, vout, win, idx_final);
        return;
    }
    float ov;
    int op, dst;
    bool has;
    wave_select_set<VPL, LAYOUT>(sv, sp, keep, KCin * KCin, scratch, has, dst, ov, op);
    if (has) {
        const long o = (b * (N >> vout) + gout) * keep + dst;
        L.pos[vout][2 * o] = (uint8_t)(op / KCin);
        L.pos[vout][2 * o + 1] = (uint8_t)(op % KCin);
        L.S[vout][o] = ov;
    }
}

// ------------------------------------------------------- combine of level 0
// Siblings n = 2g, m = 2g + 1 (single codebooks): scores straight from the leaf table.  One wave per (b, g).
template <int KC, typename CT = uint8_t>
__global__ void __launch_bounds__(64)
k_tf_pair0(const float *__restrict__ G, const CT *__restrict__ idx, const float *__restrict__ E, TfLists L, long B,
           int N, int K, int keep, CT *__restrict__ idx_final, const int *__restrict__ nact) {
    constexpr int VPL = KC * KC / 64;
    __shared__ u64 scratch[kSelectLdsU64];
    Stamps stp;
    if (nact) B = *nact;
    const int Gout = N >> 1;
    const int g = (int)(blockIdx.x & (unsigned)(Gout - 1));
    const long b = (long)(blockIdx.x >> __builtin_ctz((unsigned)Gout));
    if (b >= B) return;
    const int lane = lane_id();
    const int n = 2 * g, m = n + 1;
    const CT *en = reinterpret_cast<const CT *>(L.ent) + (b * N + n) * KC, *em = reinterpret_cast<const CT *>(L.ent) + (b * N + m) * KC;
    const int i = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
    const float Eb = E[b];
    const float se = L.S[0][(b * N + n) * KC + i];
    float so[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) so[v] = L.S[0][(b * N + m) * KC + j0 + v];
    float d[VPL];
    stp.at(1);                                  // lists and scores in
    tf_leaf<KC, CT>(G, N * K, K, n, m, en, em, idx[b * N + n], idx[b * N + m], d);
    float sv[VPL];
    int sp[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        sv[v] = ((se + so[v]) - Eb) + 2.0f * d[v];
        sp[v] = VPL * lane + v;
    }
    stp.at(2);                                  // leaf table gathered, scores formed
    tf_finish<VPL, CT>(sv, sp, keep, KC, scratch, L, 1, b, N, g, idx_final);
    stp.at(3);                                  // selection done, list written
    stp.flush(1, blockIdx.x, 64);
}

// Level 0, lists of 16 one-byte entries, in the SLOT-MAJOR layout (round 6): slot v of lane l is the pair (row i = 4 v + l / 16,
// column j = l % 16), position 64 v + l.  k_tf_pair0 above gives a lane one row and four columns, so a 16-lane group of a gather
// touches four rows x four columns -- sixteen 64-byte pieces, sixteen L1 accesses (the L1 counts one access per 16-lane group and
// 64-byte piece, and k_tf_pair0 runs at 0.89 accesses per clock and CU of the one it has: LAB_NOTEBOOK.md).  Here a group is ONE row
// and its sixteen columns: the columns of a shortlist fall into about ten of the row's sixteen pieces, so a gather costs about 41
// accesses instead of 62.  The rest follows: the row entries of a slot are the four bytes of one dword of the list (a wave-uniform
// load, the lane takes byte l / 16), the scores of a list reach the lanes as one coalesced load and four ds_bpermute, the border
// lanes reuse the column entry they hold.  Same expressions, same values; the list leaves in ascending position as ever
// (wave_select_set<kSlotMajor>).
__global__ void __launch_bounds__(64)
k_tf_pair0s(const float *__restrict__ G, const uint8_t *__restrict__ idx, const float *__restrict__ E, TfLists L, long B,
            int N, int K, int keep, uint8_t *__restrict__ idx_final, const int *__restrict__ nact) {
    constexpr int KC = 16, VPL = 4;
    __shared__ u64 scratch[kSelectLdsU64];
    Stamps stp;
    if (nact) B = *nact;
    const int Gout = N >> 1;
    const int g = (int)(blockIdx.x & (unsigned)(Gout - 1));
    const long b = (long)(blockIdx.x >> __builtin_ctz((unsigned)Gout));
    if (b >= B) return;
    const int lane = lane_id();
    const int n = 2 * g, m = n + 1;
    const int q4 = lane >> 4, c = lane & 15;
    const uint8_t *en = L.ent + (b * N + n) * KC, *em = en + KC;
    const float *Sn = L.S[0] + (b * N + n) * KC, *Sm = Sn + KC;
    // everything a lane needs from the lists, requested together
    const uint32_t *enw = reinterpret_cast<const uint32_t *>(en);
    uint32_t rw[VPL];                                        // dword v of list n = the entries of rows 4 v .. 4 v + 3
#pragma unroll
    for (int v = 0; v < VPL; ++v) rw[v] = enw[v];
    int e_j = em[c];                                         // column entry (also the border entry of lanes 16 .. 31)
    int e_b = en[c];                                         // row entry at position c: the border entry of lanes 0 .. 15
    int old_n = idx[b * N + n], old_m = idx[b * N + m];
    const float sn_c = Sn[c], so = Sm[c];
    const float Eb = E[b];
    asm volatile("" : "+v"(e_j), "+v"(e_b), "+v"(old_n), "+v"(old_m));
    stp.at(1);                                  // lists and scores in
    const int nksh = __builtin_ctz((unsigned)(N * K));
    const uint32_t rown = (uint32_t)(n * K), colm = (uint32_t)(m * K);
    float gq[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const uint32_t e_i = (rw[v] >> (8 * q4)) & 0xffu;
        gq[v] = G[((rown + e_i) << nksh) + colm + (uint32_t)e_j];
    }
    // border: lanes [0, 16) G[s_n,l][o_m], [16, 32) G[o_n][s_m,l-16], the others G[o_n][o_m]
    const uint32_t br = lane < KC ? rown + (uint32_t)e_b : rown + (uint32_t)old_n;
    const uint32_t bc = (lane >= KC && lane < 2 * KC) ? colm + (uint32_t)e_j : colm + (uint32_t)old_m;
    const float bv = G[(br << nksh) + bc];
    const float vj = shfl_f(bv, KC + c), w = shfl_f(bv, 2 * KC);
    float sv[VPL];
    int sp[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int i = 4 * v + q4;
        const float u = shfl_f(bv, i), se = shfl_f(sn_c, i);
        const float d = ((gq[v] - u) - vj) + w;
        sv[v] = ((se + so) - Eb) + 2.0f * d;
        sp[v] = 64 * v + lane;
    }
    stp.at(2);                                  // leaf table gathered, scores formed
    tf_finish<VPL, uint8_t, kSlotMajor>(sv, sp, keep, KC, scratch, L, 1, b, N, g, idx_final);
    stp.at(3);                                  // selection done, list written
    stp.flush(1, blockIdx.x, 64);
}

// ------------------------------------------------------------ level-1 tables
// T_1[X][Y] of two groups of two codebooks (X < Y) over their lists of KC candidates, each entry the sum of four
// leaf-table entries.  Lane holds positions p = VPL*lane + v.
//
// Lists of 16 (K >= 32): only the leaf entries the candidates USE are read.  A list's 16 candidates are pairs of
// positions in the two halves' level-0 lists, and on average only 7.5 of the 16 positions of a half occur
// (measured on the bench workload), so a leaf table is needed on a |A| x |B| sub-grid (56 entries instead of 256,
// plus the border).  The sets come from a 16-lane OR-reduction of position bits, the compact rank of a position is
// a popcount, and the four compact leaf tables live in LDS.  Every entry is computed by the same formula as in the
// full table, so the results are identical.
template <int KCH, int KC, typename CT = uint8_t>
__device__ __forceinline__ void tf_table1(const float *__restrict__ G, const CT *__restrict__ idx, const TfLists &L,
                                          long b, int N, int K, int X, int Y, float *leaf /* LDS [4][KCH*KCH + 64] */,
                                          float (&t)[KC * KC / 64], Stamps &stp) {
    constexpr int VPLH = KCH * KCH / 64, VPL = KC * KC / 64, MH = KCH * KCH;
    const int lane = lane_id();
    const CT *id = idx + b * N;
    const CT *ent = reinterpret_cast<const CT *>(L.ent);
    const int G1 = N >> 1;
    const uint8_t *px = L.pos[1] + ((b * G1 + X) * KC) * 2, *py = L.pos[1] + ((b * G1 + Y) * KC) * 2;
    if constexpr (KCH == 16 && KC == 16) {
        // per leaf table: 16 rows of 17 floats (an odd row stride: the quarter-wave reads below -- up to sixteen rows, a few
        // columns -- spread over all 32 banks; with rows of 16 they met in 2 x 16, SQ_LDS_BANK_CONFLICT 26 M cycles per launch)
        // + 33 border values
        constexpr int RS = KCH + 1, BO = KCH * RS;
        constexpr int LS = BO + 36;                           // (33 border values; 5,056 bytes per wave in all: 32 waves per CU)
        CT *cent = reinterpret_cast<CT *>(leaf + 4 * LS);             // [4][16] compact list -> codebook entry
        uint8_t *crank = reinterpret_cast<uint8_t *>(cent + 64);      // [4][16] candidate -> compact rank of its leaf
        const int NK = N * K;
        const int nksh = __builtin_ctz((unsigned)NK);
        // quarter w of the wave = one of the four halves' lists: 0, 1 = the halves of X, 2, 3 = those of Y
        const int w = lane >> 4, c = lane & 15;
        const int cb = (w < 2) ? 2 * X + w : 2 * Y + (w - 2);      // this quarter's codebook
        // every byte of the lists and of the current indexes this wave needs is requested here, together: left to the
        // compiler they were loaded one at a time, each behind a full wait, and the four border reads below (which only
        // need the current entries of their codebooks) were serialised behind them -- eleven round trips instead of three
        int mypos = (w < 2 ? px : py)[2 * c + (w & 1)];
        int myent = ent[(b * N + cb) * KCH + c];                    // entry at level-0 position c of that codebook
        int oldv = id[cb];                                          // current entry of this quarter's codebook
        asm volatile("" : "+v"(mypos), "+v"(myent), "+v"(oldv));
        stp.at(1);                                                  // list bytes in
        int oldq[4];                                                // current entries of codebooks 2X, 2X+1, 2Y, 2Y+1
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) oldq[qq] = __builtin_amdgcn_readlane(oldv, 16 * qq);
        uint32_t m = 1u << mypos;
        m |= (uint32_t)dpp_i<0x121>((int)m);     // row_ror 1, 2, 4, 8: OR over the 16 lanes of the quarter
        m |= (uint32_t)dpp_i<0x122>((int)m);
        m |= (uint32_t)dpp_i<0x124>((int)m);
        m |= (uint32_t)dpp_i<0x128>((int)m);
        crank[lane] = (uint8_t)__popc(m & ((1u << mypos) - 1u));
        if ((m >> c) & 1u) cent[w * 16 + __popc(m & ((1u << c) - 1u))] = (CT)myent;
        uint32_t mq[4];
        mq[0] = (uint32_t)__builtin_amdgcn_readlane((int)m, 0);
        mq[1] = (uint32_t)__builtin_amdgcn_readlane((int)m, 16);
        mq[2] = (uint32_t)__builtin_amdgcn_readlane((int)m, 32);
        mq[3] = (uint32_t)__builtin_amdgcn_readlane((int)m, 48);
        wave_lds_fence();
        // All four tables move together: their border reads are one batch of loads, their core reads another, so a
        // wave waits for two L2 round trips instead of eight (the kernel is bound by that latency chain, not by bytes).
        int na_[4], nc_[4];
        uint32_t rown_[4], colm_[4];
        float bv[4];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const int a = tb >> 1, cc = tb & 1;
            const int n = 2 * X + a, mcb = 2 * Y + cc;
            na_[tb] = __popc(mq[a]);
            nc_[tb] = __popc(mq[2 + cc]);
            rown_[tb] = (uint32_t)(n * K);
            colm_[tb] = (uint32_t)(mcb * K);
            // border: lanes [0, na) G[s_i][o_m], [na, na + nc) G[o_n][s_j], the others G[o_n][o_m]
            const bool isu = lane < na_[tb], isv = lane >= na_[tb] && lane < na_[tb] + nc_[tb];
            const uint32_t br = rown_[tb] + (uint32_t)(isu ? cent[a * 16 + lane] : oldq[a]);
            const uint32_t bc = colm_[tb] + (uint32_t)(isv ? cent[(2 + cc) * 16 + (lane - na_[tb])] : oldq[2 + cc]);
            // (G is symmetric bit for bit: the border entries G[s_i][o_m] are read as G[o_m][s_i] -- ONE row segment, about five
            // 128-byte lines, instead of a line in each of the |A| rows: 74.9 -> 72.0 M L2 requests per launch, 375 -> 369 us)
            bv[tb] = G[isu ? (bc << nksh) + br : (br << nksh) + bc];
        }
        float g[4][4];
        int rr[4][4];      // ra * 16 + rc of this lane's entry, -1 if none
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const int a = tb >> 1, cc = tb & 1;
            const int na = na_[tb], nc = nc_[tb];
            const float rnc = 1.0f / (float)nc;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = lane + 64 * it;
                rr[tb][it] = -1;
                g[tb][it] = 0.f;
                if (64 * it < na * nc) {          // wave-uniform
                    const int qc = q < na * nc ? q : 0;
                    int ra = (int)((float)qc * rnc);               // qc / nc for qc < 256 (corrected below)
                    ra -= (ra * nc > qc);
                    ra += ((ra + 1) * nc <= qc);
                    const int rc = qc - ra * nc;
                    g[tb][it] = G[((rown_[tb] + (uint32_t)cent[a * 16 + ra]) << nksh) + colm_[tb] + (uint32_t)cent[(2 + cc) * 16 + rc]];
                    rr[tb][it] = q < na * nc ? ra * 16 + rc : -1;
                }
            }
        }
        stp.at(2);                                                  // masks exchanged, every Gram gather back
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
            if (lane <= na_[tb] + nc_[tb]) leaf[tb * LS + BO + lane] = bv[tb];
        wave_lds_fence();
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            float *lt = leaf + tb * LS;
            const float wv = lt[BO + na_[tb] + nc_[tb]];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int e = rr[tb][it];
                if (e >= 0) lt[e + (e >> 4)] = ((g[tb][it] - lt[BO + (e >> 4)]) - lt[BO + na_[tb] + (e & 15)]) + wv;      // row ra at 17 ra
            }
        }
        wave_lds_fence();
        const int i = lane >> 2, j0 = 4 * (lane & 3);
        const int ri0 = crank[i], ri1 = crank[16 + i];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int rj0 = crank[32 + j0 + v], rj1 = crank[48 + j0 + v];
            t[v] = ((leaf[ri0 * RS + rj0] + leaf[LS + ri0 * RS + rj1]) + leaf[2 * LS + ri1 * RS + rj0]) +
                   leaf[3 * LS + ri1 * RS + rj1];
        }
        stp.at(3);                                                  // leaf tables assembled in LDS, level-1 entries summed
        return;
    } else if constexpr (KCH * KCH == 64) {
        // lists of 8 (16-entry codebooks): a leaf table is one core entry and one border entry per lane.  The four tables
        // move together -- all list / index bytes in one batch, all eight Gram reads in a second one -- instead of four
        // tf_leaf calls in a row (eight dependent round trips: this path is the trainer's first phase)
        const int NK = N * K;
        const int nksh = __builtin_ctz((unsigned)NK);
        const int i = lane / KCH, j = lane % KCH;
        const int bl = lane < 2 * KCH ? lane : 2 * KCH;
        int cbk[4] = {2 * X, 2 * X + 1, 2 * Y, 2 * Y + 1};
        int e_row[2], e_col[2], e_brd[4], oldq[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            e_row[a] = ent[(b * N + cbk[a]) * KCH + i];
            e_col[a] = ent[(b * N + cbk[2 + a]) * KCH + j];
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            oldq[q4] = id[cbk[q4]];
            // border entry of this lane in a table whose row codebook is cbk[a] / column codebook cbk[2 + c]: lanes [0, KCH)
            // need ent of the ROW codebook at position lane, lanes [KCH, 2 KCH) ent of the COLUMN codebook at lane - KCH
            e_brd[q4] = ent[(b * N + cbk[q4]) * KCH + (q4 < 2 ? (bl < KCH ? bl : 0) : (bl >= KCH && bl < 2 * KCH ? bl - KCH : 0))];
        }
        asm volatile("" : "+v"(e_row[0]), "+v"(e_row[1]), "+v"(e_col[0]), "+v"(e_col[1]));
        asm volatile("" : "+v"(e_brd[0]), "+v"(e_brd[1]), "+v"(e_brd[2]), "+v"(e_brd[3]));
        float g[4], bv[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t rown = (uint32_t)(cbk[a] * K), colm = (uint32_t)(cbk[2 + c] * K);
                g[a * 2 + c] = G[((rown + (uint32_t)e_row[a]) << nksh) + colm + (uint32_t)e_col[c]];
                const uint32_t br = bl < KCH ? rown + (uint32_t)e_brd[a] : rown + (uint32_t)oldq[a];
                const uint32_t bc = (bl >= KCH && bl < 2 * KCH) ? colm + (uint32_t)e_brd[2 + c] : colm + (uint32_t)oldq[2 + c];
                bv[a * 2 + c] = G[(br << nksh) + bc];
            }
        constexpr int RS = KCH + 1, TS = KCH * RS;      // rows of 9 floats (odd stride: see the lists of 16)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const float u = shfl_f(bv[tb], i), w = shfl_f(bv[tb], 2 * KCH), vj = shfl_f(bv[tb], KCH + j);
            leaf[tb * TS + i * RS + j] = ((g[tb] - u) - vj) + w;          // the expression of tf_leaf
        }
        wave_lds_fence();
        const int ii = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
        const int i0 = px[2 * ii], i1 = px[2 * ii + 1];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int jj0 = py[2 * (j0 + v)], jj1 = py[2 * (j0 + v) + 1];
            t[v] = ((leaf[i0 * RS + jj0] + leaf[TS + i0 * RS + jj1]) + leaf[2 * TS + i1 * RS + jj0]) +
                   leaf[3 * TS + i1 * RS + jj1];
        }
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n = 2 * X + a, m = 2 * Y + c;
                float d[VPLH];
                tf_leaf<KCH, CT>(G, N * K, K, n, m, ent + (b * N + n) * KCH, ent + (b * N + m) * KCH, id[n], id[m], d);
                float *dst = leaf + (a * 2 + c) * MH + VPLH * lane;
#pragma unroll
                for (int v = 0; v < VPLH; ++v) dst[v] = d[v];
            }
        wave_lds_fence();
        const int i = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
        const int i0 = px[2 * i], i1 = px[2 * i + 1];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int jj0 = py[2 * (j0 + v)], jj1 = py[2 * (j0 + v) + 1];
            t[v] = ((leaf[i0 * KCH + jj0] + leaf[MH + i0 * KCH + jj1]) + leaf[2 * MH + i1 * KCH + jj0]) +
                   leaf[3 * MH + i1 * KCH + jj1];
        }
    }
}


constexpr int tf_leaf_lds_floats(int KCH, int code_bytes = 1) { return 4 * (KCH * (KCH + 1) + 36) + 16 + 16 * code_bytes; }

// combine of level 1: siblings X = 2g, Y = 2g + 1 (pairs of codebooks).  One wave per (b, g).
template <int KCH, int KC, typename CT = uint8_t>
__device__ __forceinline__ void tf_pair1_body(unsigned bid, float *leaf, const float *__restrict__ G, const CT *__restrict__ idx,
                                              const float *__restrict__ E, const TfLists &L, long B, int N, int K, int keep,
                                              CT *__restrict__ idx_final, const int *__restrict__ nact) {
    constexpr int VPL = KC * KC / 64;
    // the selection's scratch reuses the leaf tables' LDS (dead once the level-1 table sits in registers): 5.6 instead of
    // 7.3 KB per single-wave workgroup, i.e. 29 instead of 22 waves per CU
    static_assert(tf_leaf_lds_floats(KCH, sizeof(CT)) * 4 >= kSelectLdsU64 * 8, "");
    u64 *scratch = reinterpret_cast<u64 *>(leaf);
    Stamps stp;
    if (nact) B = *nact;
    const int Gout = N >> 2;
    const int g = (int)(bid & (unsigned)(Gout - 1));
    const long b = (long)(bid >> __builtin_ctz((unsigned)Gout));
    if (b >= B) return;
    const int lane = lane_id();
    const int X = 2 * g, Y = X + 1, G1 = N >> 1;
    const int i = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
    const float Eb = E[b];
    const float se = L.S[1][(b * G1 + X) * KC + i];
    float so[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) so[v] = L.S[1][(b * G1 + Y) * KC + j0 + v];
    float t[VPL];
    tf_table1<KCH, KC, CT>(G, idx, L, b, N, K, X, Y, leaf, t, stp);
    float sv[VPL];
    int sp[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        sv[v] = ((se + so[v]) - Eb) + 2.0f * t[v];
        sp[v] = VPL * lane + v;
    }
    wave_lds_fence();                                   // the table reads are done before the selection writes there
    tf_finish<VPL, CT>(sv, sp, keep, KC, scratch, L, 2, b, N, g, idx_final);
    stp.at(4);                                          // selection done, list written
    stp.flush(2, bid, 64);
}

template <int KCH, int KC, typename CT = uint8_t>
__global__ void __launch_bounds__(64)
k_tf_pair1(const float *__restrict__ G, const CT *__restrict__ idx, const float *__restrict__ E, TfLists L, long B,
           int N, int K, int keep, CT *__restrict__ idx_final, const int *__restrict__ nact) {
    __shared__ __attribute__((aligned(16))) float leaf[tf_leaf_lds_floats(KCH, sizeof(CT))];
    tf_pair1_body<KCH, KC, CT>(blockIdx.x, leaf, G, idx, E, L, B, N, K, keep, idx_final, nact);
}

// T_1 of COUSIN pairs under the siblings of a higher level -> tabs[b][t][KC*KC].  One wave per (b, t); workgroup id
// mod ntab = t, so an XCD reads the leaf blocks of its own tables only.  Under sibling pair g each side has `per`
// level-1 groups: t = (g * per + a) * per + c  ->  X = 2 g per + a,  Y = (2 g + 1) per + c.
template <int KCH, int KC, typename CT = uint8_t>
__device__ __forceinline__ void tf_table1_body(unsigned bid, float *leaf, const float *__restrict__ G, const CT *__restrict__ idx,
                                               const TfLists &L, long B, int N, int K, int ntab, int per, float *__restrict__ tabs,
                                               const int *__restrict__ nact) {
    constexpr int VPL = KC * KC / 64;
    Stamps stp;
    if (nact) B = *nact;
    const int t = (int)(bid & (unsigned)(ntab - 1));
    const long b = (long)(bid >> __builtin_ctz((unsigned)ntab));
    if (b >= B) return;
    const int lane = lane_id();
    const int psh = __builtin_ctz((unsigned)per);
    const int c = t & (per - 1), a = (t >> psh) & (per - 1), g = t >> (2 * psh);
    const int X = 2 * g * per + a, Y = (2 * g + 1) * per + c;
    float tv[VPL];
    tf_table1<KCH, KC, CT>(G, idx, L, b, N, K, X, Y, leaf, tv, stp);
    float *dst = tabs + ((size_t)(b * ntab + t) * (KC * KC) + VPL * lane);
    if constexpr (VPL == 4) {
        __builtin_nontemporal_store((f32x4){tv[0], tv[1], tv[2], tv[3]}, reinterpret_cast<f32x4 *>(dst));
    } else {
#pragma unroll
        for (int v = 0; v < VPL; ++v) dst[v] = tv[v];
    }
    stp.at(4);                                          // table stored
    stp.flush(3, bid, 64);
}

template <int KCH, int KC, typename CT = uint8_t>
__global__ void __launch_bounds__(64)
k_tf_table1(const float *__restrict__ G, const CT *__restrict__ idx, TfLists L, long B, int N, int K, int ntab,
            int per, float *__restrict__ tabs, const int *__restrict__ nact) {
    __shared__ __attribute__((aligned(16))) float leaf[tf_leaf_lds_floats(KCH, sizeof(CT))];
    tf_table1_body<KCH, KC, CT>(blockIdx.x, leaf, G, idx, L, B, N, K, ntab, per, tabs, nact);
}

// The sibling combines of level 1 and the cousin tables the level-2 combine needs, in ONE launch: both read the level-1 lists
// and nothing of each other; the workgroup-to-XCD mapping of both kinds is what it is in their own launches (the pair
// blocks are a multiple of 8).
template <int KCH, int KC, typename CT = uint8_t>
__global__ void __launch_bounds__(64)
k_tf_level1(const float *__restrict__ G, const CT *__restrict__ idx, const float *__restrict__ E, TfLists L, long B, int N,
            int K, int keep, int ntab, int per, float *__restrict__ tabs, const int *__restrict__ nact, unsigned pair_blocks,
            unsigned tab_blocks, int ntab2, int per2, float *__restrict__ tabs2) {
    __shared__ __attribute__((aligned(16))) float leaf[tf_leaf_lds_floats(KCH, sizeof(CT))];
    if (blockIdx.x < pair_blocks) tf_pair1_body<KCH, KC, CT>(blockIdx.x, leaf, G, idx, E, L, B, N, K, keep, nullptr, nact);
    else if (blockIdx.x < pair_blocks + tab_blocks) tf_table1_body<KCH, KC, CT>(blockIdx.x - pair_blocks, leaf, G, idx, L, B, N, K, ntab, per, tabs, nact);
    else tf_table1_body<KCH, KC, CT>(blockIdx.x - pair_blocks - tab_blocks, leaf, G, idx, L, B, N, K, ntab2, per2, tabs2, nact);   // (16 codebooks: the tables of level 3 as well)
}

// copy COUNT tables of KH x KH floats each from global memory (rows of KH) to LDS (rows of KH + 1: the odd row stride keeps the
// score loops' reads -- many rows, few columns per instruction -- off each other's banks; k_tf_comb<16,32> spent 5.4 extra LDS
// cycles per read on them with rows of 16).  Wave-cooperative; the loads go out in batches of up to eight per lane before the
// first LDS write: written as a plain loop the compiler issued load, wait, write per iteration -- a memory round trip per 1 KB.
template <int KH>
constexpr int tf_tab_stride() { return KH * (KH + 1); }      // floats per table in LDS

template <int KH, int COUNT>
__device__ __forceinline__ void tf_load_tables(const float *__restrict__ src, float *dst) {
    constexpr int M = KH * KH, RS = KH + 1, TS = tf_tab_stride<KH>();
    const int lane = lane_id();
    if constexpr ((COUNT * M) % 256 == 0 && KH % 4 == 0) {
        constexpr int IT = COUNT * M / 256;               // float4 per lane
        constexpr int BATCH = IT < 8 ? IT : 8;
#pragma unroll
        for (int i0 = 0; i0 < IT; i0 += BATCH) {
            f32x4 r[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i)
                if (i0 + i < IT) r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src) + lane + 64 * (i0 + i));      // (read once: past the L2's Gram blocks)
#pragma unroll
            for (int i = 0; i < BATCH; ++i)
                if (i0 + i < IT) {
                    const int e = 4 * (lane + 64 * (i0 + i));      // element of the COUNT tables: table e / M, row, four columns
                    float *d = dst + (e / M) * TS + ((e % M) / KH) * RS + (e % KH);
#pragma unroll
                    for (int c = 0; c < 4; ++c) d[c] = r[i][c];
                }
        }
    } else {
        constexpr int TOT = COUNT * M;                    // (64-entry tables: K == 16)
        constexpr int IT = (TOT + 63) / 64;
        float r[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) r[i] = (lane + 64 * i < TOT) ? src[lane + 64 * i] : 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = lane + 64 * i;
            if (e < TOT) dst[(e / M) * TS + ((e % M) / KH) * RS + (e % KH)] = r[i];
        }
    }
}

// T of a pair of groups over lists of KC candidates from the four tables of their halves (KH x KH each, in LDS):
// lane holds positions p = VPL*lane + v.  px / py: the groups' position lists.
template <int KH, int KC>
__device__ __forceinline__ void tf_up(const float *t00, const float *t01, const float *t10, const float *t11,
                                      const uint8_t *__restrict__ px, const uint8_t *__restrict__ py,
                                      float (&t)[KC * KC / 64]) {
    constexpr int VPL = KC * KC / 64;
    const int lane = lane_id();
    const int i = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
    const int i0 = px[2 * i], i1 = px[2 * i + 1];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int jj0 = py[2 * (j0 + v)], jj1 = py[2 * (j0 + v) + 1];
        t[v] = ((t00[i0 * (KH + 1) + jj0] + t01[i0 * (KH + 1) + jj1]) + t10[i1 * (KH + 1) + jj0]) + t11[i1 * (KH + 1) + jj1];      // (rows of KH + 1: tf_load_tables)
    }
}

// ------------------------------------------------- tables of levels >= 2
// T_u of the cousin pair (X, Y) of level-u groups from the four level-(u-1) tables of their halves (tabs_in, rows of
// 2 * per tables) -> tabs_out.  Same indexing as k_tf_table1: t = (g * per + a) * per + c.  One wave per (b, t).
template <int KH, int KC>
__global__ void __launch_bounds__(64)
k_tf_up(TfLists L, long B, int N, int u, int ntab, int per, const float *__restrict__ tabs_in, float *__restrict__ tabs_out,
        const int *__restrict__ nact) {
    constexpr int VPL = KC * KC / 64, MH = KH * KH, TS = tf_tab_stride<KH>();
    __shared__ __attribute__((aligned(16))) float th[4 * TS];
    if (nact) B = *nact;
    const int t = (int)(blockIdx.x & (unsigned)(ntab - 1));
    const long b = (long)(blockIdx.x >> __builtin_ctz((unsigned)ntab));
    if (b >= B) return;
    const int lane = lane_id();
    const int psh = __builtin_ctz((unsigned)per);
    const int c = t & (per - 1), a = (t >> psh) & (per - 1), g = t >> (2 * psh);
    const int X = 2 * g * per + a, Y = (2 * g + 1) * per + c;
    const int Gu = N >> u;
    // children: tables (2a + i, 2c + j) of sibling pair g at the level below (2 * per tables per row)
    const size_t cbase = (size_t)b * (4 * ntab) + (size_t)g * (4 * per * per);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            tf_load_tables<KH, 1>(tabs_in + (cbase + (size_t)(2 * a + i) * (2 * per) + (2 * c + j)) * MH, th + (2 * i + j) * TS);
    wave_lds_fence();
    float tv[VPL];
    tf_up<KH, KC>(th, th + TS, th + 2 * TS, th + 3 * TS, L.pos[u] + ((b * Gu + X) * KC) * 2, L.pos[u] + ((b * Gu + Y) * KC) * 2, tv);
    float *dst = tabs_out + ((size_t)(b * ntab + t) * (KC * KC) + VPL * lane);
#pragma unroll
    for (int v = 0; v < VPL; ++v) __builtin_nontemporal_store(tv[v], dst + v);      // (a stream: past the L2, like the level-1 tables)
}

// ------------------------------------------------ combine of a level >= 2
// Siblings P = 2h, Q = 2h + 1 of level v (lists of KC): the four level-(v-1) tables of their halves come from tabs
// ([b][h][2][2][KH*KH]).  One wave per (b, h).  KC == 64 (4,096 pairs: only ever the last combine) streams the scores
// through a running arg-min instead of holding them.
// LAST: the combine that leaves one group (idx_final != nullptr, keep == 1) as an instantiation of its own -- the winner is a
// running arg-min over the scores, no score array and no selection: 86 registers -> the 8 waves per SIMD of the other pass kernels
// (k_tf_comb<16,32> is the last launch of every pass of 8 codebooks).
template <int KH, int KC, bool LAST, typename CT = uint8_t>
__global__ void __launch_bounds__(64)
k_tf_comb(const float *__restrict__ E, TfLists L, long B, int N, int v, int keep, const float *__restrict__ tabs,
          CT *__restrict__ idx_final, const int *__restrict__ nact) {
    constexpr int MH = KH * KH, RS = KH + 1, TS = tf_tab_stride<KH>();
    constexpr int VPL = (KC * KC / 64 <= 16) ? KC * KC / 64 : 16;
    constexpr int CHUNKS = KC * KC / (64 * VPL);
    // (the selection's scratch reuses the tables' LDS: see k_tf_pair1)
    constexpr int LDSF = (4 * TS * 4 >= kSelectLdsU64 * 8) ? 4 * TS : kSelectLdsU64 * 2;
    __shared__ __attribute__((aligned(16))) float th[LDSF];
    u64 *scratch = reinterpret_cast<u64 *>(th);
    // (the chunked selection of 64 x 64 pairs runs while the tables are still being read: its scratch is its own)
    __shared__ u64 sel2_store[(KC * KC / 64 > 16) ? kSelectLdsU64 : 1];
    u64 *sel2 = sel2_store;
    if (nact) B = *nact;
    const int Gout = N >> (v + 1);
    const int h = (int)(blockIdx.x & (unsigned)(Gout - 1));
    const long b = (long)(blockIdx.x >> __builtin_ctz((unsigned)Gout));
    if (b >= B) return;
    const int lane = lane_id();
    const int P = 2 * h, Q = P + 1, Gv = N >> v;
    tf_load_tables<KH, 4>(tabs + ((size_t)b * Gout + h) * 4 * MH, th);
    const float Eb = E[b];
    const uint8_t *px = L.pos[v] + ((b * Gv + P) * KC) * 2, *py = L.pos[v] + ((b * Gv + Q) * KC) * 2;
    const float *Sx = L.S[v] + (b * Gv + P) * KC, *Sy = L.S[v] + (b * Gv + Q) * KC;
    wave_lds_fence();
    if constexpr (CHUNKS == 1 && !LAST) {
        const int i = (VPL * lane) / KC, j0 = (VPL * lane) % KC;
        const float se = Sx[i];
        float t[VPL];
        tf_up<KH, KC>(th, th + TS, th + 2 * TS, th + 3 * TS, px, py, t);
        float sv[VPL];
        int sp[VPL];
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            sv[u] = ((se + Sy[j0 + u]) - Eb) + 2.0f * t[u];
            sp[u] = VPL * lane + u;
        }
        wave_lds_fence();
        tf_finish<VPL, CT>(sv, sp, keep, KC, scratch, L, v + 1, b, N, h, idx_final);
    } else if (!LAST) {
        // 4,096 pairs that are NOT the last combine (64 codebooks of more than 16 entries: 64 of them go on): the 64 smallest
        // of every chunk of 1,024, then the 64 smallest of those 4 x 64 -- the same set and order as one selection over
        // all of them (keys are unique)
        static_assert(CHUNKS <= 4, "");
        float cv[4];
        int cp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { cv[c] = INFINITY; cp[c] = kBigPos; }
        for (int c = 0; c < CHUNKS; ++c) {
            const int p0 = 64 * VPL * c + VPL * lane;
            const int i = p0 / KC, j0 = p0 % KC;
            const int i0 = px[2 * i], i1 = px[2 * i + 1];
            const float se = Sx[i];
            float sv[VPL];
            int sp[VPL];
#pragma unroll
            for (int u = 0; u < VPL; ++u) {
                const int jj0 = py[2 * (j0 + u)], jj1 = py[2 * (j0 + u) + 1];
                const float t = ((th[i0 * RS + jj0] + th[TS + i0 * RS + jj1]) + th[2 * TS + i1 * RS + jj0]) + th[3 * TS + i1 * RS + jj1];
                sv[u] = ((se + Sy[j0 + u]) - Eb) + 2.0f * t;
                sp[u] = p0 + u;
            }
            float ov;
            int op, dst;
            bool got;
            wave_select_set<VPL>(sv, sp, keep, KC * KC, sel2, got, dst, ov, op);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == c) { cv[q] = got ? ov : INFINITY; cp[q] = got ? op : kBigPos; }
        }
        // (the chunk survivors lie wherever their chunk's selection left them: the merge orders its list by position itself)
        tf_finish<4, CT, kAnyOrder>(cv, cp, keep, KC, sel2, L, v + 1, b, N, h, nullptr);
    } else {
        float bv = INFINITY;
        int bp = kBigPos;
        for (int c = 0; c < CHUNKS; ++c) {
            const int p0 = 64 * VPL * c + VPL * lane;             // positions p0 .. p0 + VPL - 1 share the row i
            const int i = p0 / KC, j0 = p0 % KC;
            const int i0 = px[2 * i], i1 = px[2 * i + 1];
            const float se = Sx[i];
#pragma unroll
            for (int u = 0; u < VPL; ++u) {
                const int jj0 = py[2 * (j0 + u)], jj1 = py[2 * (j0 + u) + 1];
                const float t = ((th[i0 * RS + jj0] + th[TS + i0 * RS + jj1]) + th[2 * TS + i1 * RS + jj0]) + th[3 * TS + i1 * RS + jj1];
                lexmin(bv, bp, ((se + Sy[j0 + u]) - Eb) + 2.0f * t, p0 + u);
            }
        }
        wave_lexmin(bv, bp);
        if (bp > KC * KC - 1) bp = KC * KC - 1;                  // only reachable with NaN keys
        tf_emit(L, b, N, v + 1, bp, idx_final);                   // keep == 1, last combine
    }
}

// ------------------------------------------------------- combine of level 3
// N = 16: the two groups of eight codebooks.  16 level-1 tables (tabs, quads layout) -> the four level-2 tables of
// (groups 0 | 1) x (2 | 3) in LDS -> the KC3 x KC3 scores -> their arg min; always the last combine.  One workgroup of FOUR
// waves per vector: the tables take 33 KB of LDS (16 + 16 KB at K >= 32), which leaves room for four workgroups per CU --
// with a single wave each (the first version) a CU held four waves and the kernel took 1.0 ms at 65,536 vectors.  Wave w
// loads four of the level-1 tables, builds level-2 table w and scores a quarter of the candidate pairs.
template <int KC1, int KC2, int KC3, typename CT = uint8_t>
__global__ void __launch_bounds__(256)
k_tf_comb3(const CT *__restrict__ idx, const float *__restrict__ E, TfLists L, long B, int N,
           const float *__restrict__ tabs, CT *__restrict__ idx_final, const int *__restrict__ nact) {
    constexpr int VPL2 = KC2 * KC2 / 64, M1 = KC1 * KC1;
    constexpr int PW = KC3 * KC3 / 4, VPLW = PW / 64;          // candidate pairs per wave, per lane
    constexpr int RS1 = KC1 + 1, TS1 = tf_tab_stride<KC1>(), RS2 = KC2 + 1, TS2 = KC2 * RS2;      // rows of an odd length: tf_load_tables
    __shared__ __attribute__((aligned(16))) float t1[16 * TS1];
    __shared__ __attribute__((aligned(16))) float t2[4 * TS2];
    __shared__ float wv[4];
    __shared__ int wp[4];
    (void)idx;
    if (nact) B = *nact;
    const long b = blockIdx.x;
    if (b >= B) return;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int G2 = N >> 2;   // 4 level-2 groups
    // level-2 groups xc and 2 + yc (this wave's table); their halves are the level-1 groups 2xc, 2xc+1 and 4+2yc, 4+2yc+1
    const int xc = w >> 1, yc = w & 1;
    // Every list byte and score this wave will need -- for its level-2 table and for its quarter of the candidate pairs --
    // is requested here, before the tables: read where they are used they were single-byte loads, each behind a full wait
    constexpr int NB2 = 2 * VPL2;                                 // position bytes of the lane's VPL2 columns (8 or 32)
    const int i2 = (VPL2 * lane) / KC2, j2 = (VPL2 * lane) % KC2;
    const uint8_t *px2 = L.pos[2] + ((b * G2 + xc) * KC2) * 2, *py2 = L.pos[2] + ((b * G2 + 2 + yc) * KC2) * 2;
    const unsigned pxw2 = *reinterpret_cast<const uint16_t *>(px2 + 2 * i2);
    uint32_t pyw2[NB2 / 4];
#pragma unroll
    for (int u = 0; u < NB2 / 4; ++u) pyw2[u] = reinterpret_cast<const uint32_t *>(py2 + 2 * j2)[u];
    const uint8_t *px = L.pos[3] + ((b * 2 + 0) * KC3) * 2, *py = L.pos[3] + ((b * 2 + 1) * KC3) * 2;
    const float *Sx = L.S[3] + (b * 2 + 0) * KC3, *Sy = L.S[3] + (b * 2 + 1) * KC3;
    // candidate pair p = i * KC3 + j of the two level-3 lists: wave w takes p in [w PW, (w + 1) PW), ascending per lane;
    // a lane's VPLW pairs share the row i (VPLW divides KC3)
    const int p0 = w * PW + VPLW * lane;
    const int i3 = p0 / KC3, j3 = p0 % KC3;
    const unsigned pxw3 = *reinterpret_cast<const uint16_t *>(px + 2 * i3);
    unsigned pyw3[VPLW];
    float sy3[VPLW];
#pragma unroll
    for (int v = 0; v < VPLW; ++v) {
        pyw3[v] = *reinterpret_cast<const uint16_t *>(py + 2 * (j3 + v));
        sy3[v] = Sy[j3 + v];
    }
    const float sx3 = Sx[i3];
    const float Eb = E[b];
    tf_load_tables<KC1, 4>(tabs + ((size_t)b * 16 + 4 * w) * M1, t1 + 4 * w * TS1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    {
        const int X0 = 2 * xc, Y0 = 2 * yc;      // Y0 relative to 4
        const float *t00 = t1 + (4 * X0 + Y0) * TS1, *t01 = t1 + (4 * X0 + Y0 + 1) * TS1, *t10 = t1 + (4 * (X0 + 1) + Y0) * TS1,
                    *t11 = t1 + (4 * (X0 + 1) + Y0 + 1) * TS1;
        const int i0 = (int)(pxw2 & 0xffu), i1 = (int)(pxw2 >> 8);
        float *dst = t2 + w * TS2 + i2 * RS2 + j2;          // (a lane's VPL2 columns lie in one row: VPL2 divides KC2)
#pragma unroll
        for (int v = 0; v < VPL2; ++v) {
            const uint32_t wj = pyw2[v >> 1] >> (16 * (v & 1));
            const int jj0 = (int)(wj & 0xffu), jj1 = (int)((wj >> 8) & 0xffu);
            dst[v] = ((t00[i0 * RS1 + jj0] + t01[i0 * RS1 + jj1]) + t10[i1 * RS1 + jj0]) + t11[i1 * RS1 + jj1];
        }
    }
    __syncthreads();
    float bv = INFINITY;
    int bp = kBigPos;
    {
        const int i0 = (int)(pxw3 & 0xffu), i1 = (int)(pxw3 >> 8);
#pragma unroll
        for (int v = 0; v < VPLW; ++v) {
            const int jj0 = (int)(pyw3[v] & 0xffu), jj1 = (int)(pyw3[v] >> 8);
            const float t = ((t2[i0 * RS2 + jj0] + t2[TS2 + i0 * RS2 + jj1]) + t2[2 * TS2 + i1 * RS2 + jj0]) + t2[3 * TS2 + i1 * RS2 + jj1];
            lexmin(bv, bp, ((sx3 + sy3[v]) - Eb) + 2.0f * t, p0 + v);
        }
    }
    wave_lexmin(bv, bp);
    if (lane == 0) { wv[w] = bv; wp[w] = bp; }
    __syncthreads();
    if (w == 0) {
        float rv = wv[0];
        int rp = wp[0];
#pragma unroll
        for (int u = 1; u < 4; ++u) lexmin(rv, rp, wv[u], wp[u]);
        if (rp > KC3 * KC3 - 1) rp = KC3 * KC3 - 1;              // only reachable with NaN keys
        tf_emit(L, b, N, 4, rp, idx_final);
    }
}

}  // namespace mcq
