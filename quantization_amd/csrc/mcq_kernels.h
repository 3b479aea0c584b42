// mcq_kernels.h -- gfx950 (CDNA4, wave64) kernels of the multi-codebook quantizer.
//
// Numeric contract (identical to oracle/mcq_oracle.c, which restates
// /root/reference/quantization/quantization.py:277-547):
//   * every contraction over the feature axis is ONE v_mfma_f32_16x16x4_f32
//     accumulation chain per output, accumulator starting at +0, consuming k in
//     the order  for blk: for i in 0..3: for g in 0..3: k = 16*blk + 4*g + i
//     (lane (r, g) of the wave holds the float4 at 16*blk + 4*g of row r and
//     feeds component i to MFMA number 4*blk + i of the chain);
//   * every sum of squares is 64 per-lane fmaf chains over the float4 groups
//     q = lane, lane+64, ... followed by the xor butterfly 32,16,8,4,2,1;
//   * everything else is a single IEEE fp32 operation in the reference's order;
//   * selections order candidates by (value, position), lowest position on ties.
// Compile with -ffp-contract=off: only explicit fmaf()/MFMA fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mcq {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kBigPos = 0x7fffffff;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum_butterfly(float p) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) p = p + __shfl_xor(p, m, 64);
    return p;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}

// (v, p) := min((v, p), (ov, op)) in (value, position) order
__device__ __forceinline__ void lexmin(float &v, int &p, float ov, int op) {
    const bool take = (ov < v) || (ov == v && op < p);
    v = take ? ov : v;
    p = take ? op : p;
}

// wave-wide lexicographic minimum; result uniform in every lane
__device__ __forceinline__ void wave_lexmin(float &v, int &p) {
    lexmin(v, p, dpp_f<0xB1>(v), dpp_i<0xB1>(p));    // quad_perm [1,0,3,2]  (xor 1)
    lexmin(v, p, dpp_f<0x4E>(v), dpp_i<0x4E>(p));    // quad_perm [2,3,0,1]  (xor 2)
    lexmin(v, p, dpp_f<0x141>(v), dpp_i<0x141>(p));  // row_half_mirror      (xor 7)
    lexmin(v, p, dpp_f<0x140>(v), dpp_i<0x140>(p));  // row_mirror           (xor 15)
    float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    int p0 = __builtin_amdgcn_readlane(p, 0), p1 = __builtin_amdgcn_readlane(p, 16);
    int p2 = __builtin_amdgcn_readlane(p, 32), p3 = __builtin_amdgcn_readlane(p, 48);
    lexmin(r0, p0, r1, p1);
    lexmin(r2, p2, r3, p3);
    lexmin(r0, p0, r2, p2);
    v = r0;
    p = p0;
}

// The `cnt` smallest of the wave's VPL*64 keys (v[i], p[i]) in ascending
// (value, position) order; lane j (< cnt <= 64) receives the j-th.  Mirrors
// select_smallest() of the oracle, including its treatment of non-finite keys.
template <int VPL>
__device__ __forceinline__ void wave_select(const float (&v)[VPL], const int (&p)[VPL], int cnt, int M,
                                            float &out_v, int &out_p) {
    float pv = -INFINITY;
    int pp = -1;
    const int lane = lane_id();
    out_v = INFINITY;
    out_p = M - 1;
    for (int j = 0; j < cnt; ++j) {
        float bv = INFINITY;
        int bp = kBigPos;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool gt = (v[i] > pv) || (v[i] == pv && p[i] > pp);
            const bool lt = (v[i] < bv) || (v[i] == bv && p[i] < bp);
            if (gt && lt) { bv = v[i]; bp = p[i]; }
        }
        wave_lexmin(bv, bp);
        if (bp > M - 1) bp = M - 1;  // only reachable with NaN keys
        if (lane == j) { out_v = bv; out_p = bp; }
        pv = bv;
        pp = bp;
    }
}

// ---- fast selection -----------------------------------------------------------
// Same result as wave_select() for finite scores, ~5x fewer instructions: every
// candidate becomes one unique 64-bit key (order-preserving map of the fp32 score in
// the high word, position in the low word), a ballot-driven quickselect finds the
// cnt-th smallest key T, the cnt keys <= T are compacted through LDS and ranked
// against each other.  Scores are never -0 (sums of squares and x - x are +0), so
// integer order of the keys equals (value, position) order.
typedef unsigned long long u64;
constexpr u64 kKeyMax = ~0ull;
// fp16 ingestion (MCQ_ENCODE_X_FP16): x rows are _Float16 in HBM and widen to fp32 in the load path;
// every fp16 value is exactly representable, so the codes equal those of the widened input.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_h4(const void *p) {
    return __builtin_convertvector(*reinterpret_cast<const f16x4 *>(p), f32x4);
}
constexpr int kSelectLdsU64 = 208;  // per-wave LDS scratch of wave_select_fast, in u64 (136 survivors + 64 results)

__device__ __forceinline__ uint32_t ord32(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float unord32(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}
__device__ __forceinline__ u64 readlane_u64(u64 x, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int VPL>
__device__ __forceinline__ void wave_select_fast(const float (&v)[VPL], const int (&p)[VPL], int cnt, int M,
                                                 u64 *lds /* kSelectLdsU64 per wave */, float &out_v, int &out_p) {
    if (cnt == 1) {  // plain arg-min
        wave_select<VPL>(v, p, 1, M, out_v, out_p);
        return;
    }
    const int lane = lane_id();
    u64 key[VPL];
    bool cand[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        cand[i] = p[i] != kBigPos;
        key[i] = cand[i] ? (((u64)ord32(v[i]) << 32) | (uint32_t)p[i]) : kKeyMax;
    }
    const int target = cnt - 1;
    u64 *ldsA = lds, *ldsB = lds + 64;
    if (VPL > 1 && cnt * VPL <= 64) {
        // Two-level variant (the common 256 -> 16 case).  T0 = the cnt-th smallest of the 64
        // per-lane minima bounds the answer from above: every one of the cnt smallest keys is
        // <= T0, and at most cnt lanes (x VPL keys) hold keys <= T0, so the survivors fit in one
        // key per lane; they are then ranked exactly.  The quickselect runs on one key per lane.
        u64 lmin = key[0];
#pragma unroll
        for (int i = 1; i < VPL; ++i) lmin = key[i] < lmin ? key[i] : lmin;
        // The candidate set is a wave-uniform 64-bit MASK handled by scalar instructions (a per-lane bool carried
        // round the loop is materialised as 0/1 VGPRs with v_cndmask / v_cmp pairs and nops on every round: the
        // selection is bound by exactly that scalar/VALU ping-pong).  Keys are unique, so the lanes above the
        // pivot are the complement of those below it minus the pivot's lane.
        u64 cm = __ballot(lmin != kKeyMax);
        u64 T0 = kKeyMax;
        while (cm != 0) {     // every round removes at least the pivot's lane from the candidates: <= 64 rounds
            const int pl = __ffsll((long long)cm) - 1;
            const u64 kp = readlane_u64(lmin, pl);
            const u64 ltm = __ballot(lmin < kp);
            const int rr = __popcll(ltm);
            if (rr == target) { T0 = kp; break; }
            cm &= (rr > target) ? ltm : ~(ltm | (1ull << pl));
        }
        int base = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool sel = key[i] <= T0;
            const u64 m = __ballot(sel);
            const int dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (sel && dst < 64) ldsA[dst] = key[i];
            base += __popcll(m);
        }
        const int c0 = base < 64 ? base : 64;
        // ranking walks the survivors eight at a time; the tail is padded with sentinels (never smaller than a key)
        // instead of a one-by-one remainder loop that waits out an LDS round trip per element
        if (lane >= c0 && lane < c0 + 8) ldsA[lane] = kKeyMax;      // may run into ldsB, which is written later
        wave_lds_fence();
        const u64 k = (lane < c0) ? ldsA[lane] : kKeyMax;
        int rnk = 0;
        const int c8 = (c0 + 7) & ~7;
        for (int j = 0; j < c8; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rnk += (ldsA[j + u] < k) ? 1 : 0;
        }
        if (lane < c0 && rnk < cnt) ldsB[rnk] = k;
        wave_lds_fence();
        out_v = INFINITY;
        out_p = M - 1;
        if (lane < cnt) {
            const u64 o = ldsB[lane];
            out_p = (int)(uint32_t)o;
            out_v = unord32((uint32_t)(o >> 32));
        }
        wave_lds_fence();
        return;
    }
    if (VPL > 1 && cnt * VPL <= 128 && cnt <= 64) {
        // Two-level variant with up to two survivors per lane (32 of 256 with four keys per lane): T0 as above bounds
        // the answer, at most cnt lanes hold keys <= T0, so at most cnt * VPL <= 128 keys survive; lane l ranks the
        // survivors l and l + 64 against all of them.
        u64 lmin = key[0];
#pragma unroll
        for (int i = 1; i < VPL; ++i) lmin = key[i] < lmin ? key[i] : lmin;
        u64 cm = __ballot(lmin != kKeyMax);
        u64 T0 = kKeyMax;
        while (cm != 0) {
            const int pl = __ffsll((long long)cm) - 1;
            const u64 kp = readlane_u64(lmin, pl);
            const u64 ltm = __ballot(lmin < kp);
            const int rr = __popcll(ltm);
            if (rr == target) { T0 = kp; break; }
            cm &= (rr > target) ? ltm : ~(ltm | (1ull << pl));
        }
        u64 *ldsS = lds, *ldsO = lds + 144;        // survivors [0, 136), results [144, 208)
        int base = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool sel = key[i] <= T0;
            const u64 m = __ballot(sel);
            const int dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (sel && dst < 128) ldsS[dst] = key[i];
            base += __popcll(m);
        }
        const int c0 = base < 128 ? base : 128;
        if (lane < 8) ldsS[c0 + lane] = kKeyMax;                       // sentinel tail
        wave_lds_fence();
        const u64 ka = (lane < c0) ? ldsS[lane] : kKeyMax;
        const u64 kb = (lane + 64 < c0) ? ldsS[lane + 64] : kKeyMax;
        int ra = 0, rb = 0;
        const int c8 = (c0 + 7) & ~7;
        for (int j = 0; j < c8; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const u64 o = ldsS[j + u];
                ra += (o < ka) ? 1 : 0;
                rb += (o < kb) ? 1 : 0;
            }
        }
        if (lane < c0 && ra < cnt) ldsO[ra] = ka;
        if (lane + 64 < c0 && rb < cnt) ldsO[rb] = kb;
        wave_lds_fence();
        out_v = INFINITY;
        out_p = M - 1;
        if (lane < cnt) {
            const u64 o = ldsO[lane];
            out_p = (int)(uint32_t)o;
            out_v = unord32((uint32_t)(o >> 32));
        }
        wave_lds_fence();
        return;
    }
    // quickselect: find the key T with exactly cnt - 1 keys below it.  Candidate sets as wave-uniform masks, one
    // per key slot (see the two-level variant above); the pivot is the first candidate of the lowest slot that has one.
    u64 T = 0;
    u64 cmask[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) cmask[i] = __ballot(cand[i]);
    for (;;) {                // every round removes at least the pivot from the candidates: <= 64 * VPL rounds
        u64 kp = kKeyMax;
        int ps = -1, pl = 0;
#pragma unroll
        for (int i = VPL - 1; i >= 0; --i)
            if (cmask[i] != 0) { ps = i; pl = __ffsll((long long)cmask[i]) - 1; }
        if (ps < 0) { T = kp; break; }
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (i == ps) kp = readlane_u64(key[i], pl);      // uniform branch: one slot matches
        u64 ltm[VPL];
        int r = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            ltm[i] = __ballot(key[i] < kp);
            r += __popcll(ltm[i]);
        }
        if (r == target) { T = kp; break; }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const u64 pivot_bit = (i == ps) ? (1ull << pl) : 0ull;
            cmask[i] &= (r > target) ? ltm[i] : ~(ltm[i] | pivot_bit);
        }
    }
    // compact the cnt selected keys to lds[0..cnt)
    int base = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bool sel = key[i] <= T;
        const u64 m = __ballot(sel);
        const int dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (sel && dst < 64) ldsA[dst] = key[i];
        base += __popcll(m);
    }
    if (lane >= cnt && lane < cnt + 8) ldsA[lane] = kKeyMax;       // sentinel tail: see the two-level variant
    wave_lds_fence();
    const u64 k = (lane < cnt) ? ldsA[lane] : kKeyMax;
    int rnk = 0;
    const int c8 = (cnt + 7) & ~7;
    for (int j = 0; j < c8; j += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) rnk += (ldsA[j + u] < k) ? 1 : 0;
    }
    if (lane < cnt) ldsB[rnk] = k;
    wave_lds_fence();
    out_v = INFINITY;
    out_p = M - 1;
    if (lane < cnt) {
        const u64 o = ldsB[lane];
        out_p = (int)(uint32_t)o;
        out_v = unord32((uint32_t)(o >> 32));
    }
    wave_lds_fence();
}

// ------------------------------------------------------------------- prepare
// One wave per row: dst[row][0..Dp) = scale * src[row][0..D) zero padded; Q[row] = sumsq64.
// (get_centers(), quantization.py:77-79; all_centers_sumsq, :411)
__global__ void k_prepare_rows(const float *__restrict__ src, float scale, int apply_scale, long rows, int D,
                               int Dp, float *__restrict__ dst, float *__restrict__ Q,
                               const float *__restrict__ scale_ptr /* overrides `scale` when non-null */) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (scale_ptr) scale = *scale_ptr;
    const int lane = lane_id();
    const float *s = src + row * D;
    float *d = dst + row * Dp;
    float part = 0.f;
    for (int q = lane; q < Dp / 4; q += 64) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * q + c;
            float val = (k < D) ? s[k] : 0.f;
            if (apply_scale) val = scale * val;
            o[c] = val;
            part = fmaf(val, val, part);
        }
        *reinterpret_cast<f32x4 *>(d + 4 * q) = o;
    }
    part = wave_sum_butterfly(part);
    if (Q != nullptr && lane == 0) Q[row] = part;
}

// ------------------------------------------------------------------ residual
// One wave per vector (quantization.py:338-340, :401-409):
//   xerr[b] = (old_0 + old_1 + ... ) - x[b];  E[b] = sumsq64(xerr);
//   R[b][n] = sumsq64(xerr - old_n),  old_n = C[n][idx[b][n]].
// `nact` / `map` (both nullable) serve the optional fixed-point skipping of mcq_encode_ex: the
// refinement kernels then work on a packed list of *nact still-active vectors whose rows of x are
// x[map[slot]]; every other per-vector array is indexed by slot.
__global__ void k_residual(const float *__restrict__ x, const uint8_t *__restrict__ idx,
                           const float *__restrict__ C, long B, int N, int K, int D, int Dp,
                           float *__restrict__ xerr, float *__restrict__ E, float *__restrict__ R,
                           const int *__restrict__ nact, const int *__restrict__ map, int xh /* x is fp16 */) {
    const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (nact) B = *nact;
    if (b >= B) return;
    const int lane = lane_id();
    const uint8_t *id = idx + b * N;
    const float *xb = x + (map ? (long)map[b] : b) * D;
    const _Float16 *xbh = reinterpret_cast<const _Float16 *>(x) + (map ? (long)map[b] : b) * D;
    float *xe = xerr + b * Dp;
    const bool vec_ok = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & (xh ? 7 : 15)) == 0);
    float pe = 0.f;
    for (int q = lane; q < Dp / 4; q += 64) {
        f32x4 t = *reinterpret_cast<const f32x4 *>(C + ((long)id[0]) * Dp + 4 * q);
        for (int n = 1; n < N; ++n)
            t = t + *reinterpret_cast<const f32x4 *>(C + ((long)n * K + id[n]) * Dp + 4 * q);
        f32x4 xv;
        if (vec_ok && 4 * q + 3 < D) {
            xv = xh ? load_h4(xbh + 4 * q) : *reinterpret_cast<const f32x4 *>(xb + 4 * q);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kc = (4 * q + c < D) ? 4 * q + c : 0;
                const float val = xh ? (float)xbh[kc] : xb[kc];
                xv[c] = (4 * q + c < D) ? val : 0.f;
            }
        }
        t = t - xv;
        *reinterpret_cast<f32x4 *>(xe + 4 * q) = t;
#pragma unroll
        for (int c = 0; c < 4; ++c) pe = fmaf(t[c], t[c], pe);
    }
    pe = wave_sum_butterfly(pe);
    if (lane == 0) E[b] = pe;
    for (int n = 0; n < N; ++n) {
        const float *o = C + ((long)n * K + id[n]) * Dp;
        float pr = 0.f;
        for (int q = lane; q < Dp / 4; q += 64) {
            // this lane wrote xe[4q..4q+3] above
            f32x4 t = *reinterpret_cast<const f32x4 *>(xe + 4 * q) - *reinterpret_cast<const f32x4 *>(o + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) pr = fmaf(t[c], t[c], pr);
        }
        pr = wave_sum_butterfly(pr);
        if (lane == 0) R[b * N + n] = pr;
    }
}

// Register-resident variant for the common small shapes: the N x J old-row float4s of a vector
// stay in registers between the x_err pass and the R[n] pass, so every codebook row is fetched once
// (the generic kernel re-reads them).  Same operation order, same results.
template <int NN, int J>
__global__ void k_residual_reg(const float *__restrict__ x, const uint8_t *__restrict__ idx,
                               const float *__restrict__ C, long B, int K, int D, int Dp,
                               float *__restrict__ xerr, float *__restrict__ E, float *__restrict__ R,
                               const int *__restrict__ nact, const int *__restrict__ map, int xh /* x is fp16 */) {
    const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (nact) B = *nact;
    if (b >= B) return;
    const int lane = lane_id();
    const uint8_t *id = idx + b * NN;
    const float *xb = x + (map ? (long)map[b] : b) * D;
    const _Float16 *xbh = reinterpret_cast<const _Float16 *>(x) + (map ? (long)map[b] : b) * D;
    float *xe = xerr + b * Dp;
    const bool vec_ok = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & (xh ? 7 : 15)) == 0);
    const int nq = Dp / 4;
    f32x4 rows[NN][J];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const float *o = C + ((long)n * K + id[n]) * Dp;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int q = lane + 64 * j;
            rows[n][j] = (q < nq) ? *reinterpret_cast<const f32x4 *>(o + 4 * (q < nq ? q : 0)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 xev[J];
    float pe = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int q = lane + 64 * j;
        f32x4 t = rows[0][j];
#pragma unroll
        for (int n = 1; n < NN; ++n) t = t + rows[n][j];
        f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < nq) {
            if (vec_ok && 4 * q + 3 < D) {
                xv = xh ? load_h4(xbh + 4 * q) : *reinterpret_cast<const f32x4 *>(xb + 4 * q);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kc = (4 * q + c < D) ? 4 * q + c : 0;
                    const float val = xh ? (float)xbh[kc] : xb[kc];
                    xv[c] = (4 * q + c < D) ? val : 0.f;
                }
            }
        }
        t = t - xv;
        xev[j] = t;
        if (q < nq) {
            *reinterpret_cast<f32x4 *>(xe + 4 * q) = t;
#pragma unroll
            for (int c = 0; c < 4; ++c) pe = fmaf(t[c], t[c], pe);
        }
    }
    pe = wave_sum_butterfly(pe);
    if (lane == 0) E[b] = pe;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        float pr = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (lane + 64 * j < nq) {
                const f32x4 t = xev[j] - rows[n][j];
#pragma unroll
                for (int c = 0; c < 4; ++c) pr = fmaf(t[c], t[c], pr);
            }
        }
        pr = wave_sum_butterfly(pr);
        if (lane == 0) R[b * NN + n] = pr;
    }
}

// Same, for rows too long to keep whole in registers (dim 1024 with 8 or 16 codebooks): the J float4 columns of a
// lane are walked in chunks of JC; each chunk's NN x JC row pieces stay in registers between the x_err and the R[n]
// use.  The per-lane chains of E and R[n] run over ascending q = lane + 64 j exactly as in k_residual.
template <int NN, int JC>
__global__ void k_residual_regc(const float *__restrict__ x, const uint8_t *__restrict__ idx,
                                const float *__restrict__ C, long B, int K, int D, int Dp,
                                float *__restrict__ xerr, float *__restrict__ E, float *__restrict__ R,
                                const int *__restrict__ nact, const int *__restrict__ map, int xh /* x is fp16 */) {
    const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (nact) B = *nact;
    if (b >= B) return;
    const int lane = lane_id();
    const uint8_t *id = idx + b * NN;
    const float *xb = x + (map ? (long)map[b] : b) * D;
    const _Float16 *xbh = reinterpret_cast<const _Float16 *>(x) + (map ? (long)map[b] : b) * D;
    float *xe = xerr + b * Dp;
    const bool vec_ok = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & (xh ? 7 : 15)) == 0);
    const int nq = Dp / 4;
    const float *rowp[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) rowp[n] = C + ((long)n * K + id[n]) * Dp;
    float pe = 0.f, pr[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) pr[n] = 0.f;
    for (int j0 = 0; 64 * j0 < nq; j0 += JC) {
        f32x4 rows[NN][JC];
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int j = 0; j < JC; ++j) {
                const int q = lane + 64 * (j0 + j);
                rows[n][j] = *reinterpret_cast<const f32x4 *>(rowp[n] + 4 * (q < nq ? q : 0));
            }
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const int q = lane + 64 * (j0 + j);
            if (q < nq) {
                f32x4 t = rows[0][j];
#pragma unroll
                for (int n = 1; n < NN; ++n) t = t + rows[n][j];
                f32x4 xv;
                if (vec_ok && 4 * q + 3 < D) {
                    xv = xh ? load_h4(xbh + 4 * q) : *reinterpret_cast<const f32x4 *>(xb + 4 * q);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int kc = (4 * q + c < D) ? 4 * q + c : 0;
                        const float val = xh ? (float)xbh[kc] : xb[kc];
                        xv[c] = (4 * q + c < D) ? val : 0.f;
                    }
                }
                t = t - xv;
                *reinterpret_cast<f32x4 *>(xe + 4 * q) = t;
#pragma unroll
                for (int c = 0; c < 4; ++c) pe = fmaf(t[c], t[c], pe);
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const f32x4 u = t - rows[n][j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) pr[n] = fmaf(u[c], u[c], pr[n]);
                }
            }
        }
    }
    pe = wave_sum_butterfly(pe);
    if (lane == 0) E[b] = pe;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const float v = wave_sum_butterfly(pr[n]);
        if (lane == 0) R[b * NN + n] = v;
    }
}

// ---------------------------------------------------------------------- GEMM
// out[b][n][k] = dot16(Bm[n][k][:], A_n[b][:]) for a 64-vector tile and one
// codebook n per workgroup (4 waves, wave w owns vectors 16w..16w+15 and all
// K = 16*T entries).  MFMA rows = codebook entries, columns = vectors, so the K
// scores of one vector live in the 4 lanes {c, c+16, c+32, c+48}.
//   MODE_LOGITS : A_n[b] = lscale * x[b]            (quantization.py:278)
//                 epilogue: + bias, first-max argmax over k  (:279, :301)
//   MODE_STAGE0 : A_n[b] = xerr[b] - C[n][idx[b][n]] (:403)
//                 epilogue: S = (R + Q) + 2*dot      (:418), stored to S0
//   MODE_LOGITS_OUT: as MODE_LOGITS but stores the logits (test hook)
//   MODE_STAGE0_SEL: MODE_STAGE0 + the first sort-and-truncate (:470-503) in the epilogue: the scores
//                 go through LDS to one wave per vector and only the `keep` survivors reach HBM
//                 (k_gemm8s only)
//   MODE_XC     : A_n[b] = x[b] as is; epilogue: the raw products dot16(Bm[n][k], x[b]) are stored.  Builds the
//                 table form's XC (Bm = C) and, with the scaled centers themselves as "vectors", the Gram matrix.
enum { MODE_LOGITS = 0, MODE_STAGE0 = 1, MODE_LOGITS_OUT = 2, MODE_STAGE0_SEL = 3, MODE_XC = 4 };

constexpr int kGemmVec = 64;  // vectors per workgroup
constexpr int kGemmBK = 32;   // floats of the feature axis per LDS stage (2 k-blocks)

// LDS image of a [rows][32 floats] stage: 16-byte units at
//   kb*4*rows + g*rows + (row ^ (g | kb<<2))      kb in {0,1}, g in 0..3
// conflict-free for the fragment ds_read_b128 (lane (r,g) reads row base+r, unit g)
// and for the staging ds_write_b128 (8 consecutive lanes write one row's 8 units).
__device__ __forceinline__ int lds_unit(int rows, int row, int kb, int g) {
    return kb * 4 * rows + g * rows + (row ^ (g | (kb << 2)));
}

template <int T, int MODE>
__global__ void __launch_bounds__(256, 2)
k_gemm(const float *__restrict__ Bm /*[N][K][Dp]*/, const float *__restrict__ xin /*x [B][D] or xerr [B][Dp]*/,
       const uint8_t *__restrict__ idx_in, float lscale, const float *__restrict__ bias,
       const float *__restrict__ Rin, const float *__restrict__ Qin, long B, int N, int D, int Dp,
       uint8_t *__restrict__ idx_out, float *__restrict__ out, int /*keep: k_gemm8s only*/,
       const int *__restrict__ nact, const float *__restrict__ lscale_ptr /* overrides lscale when non-null */,
       int xh /* logits modes: x is fp16 */) {
    constexpr int K = 16 * T;
    if (nact) B = *nact;
    if (lscale_ptr) lscale = *lscale_ptr;
    const int nsh = __builtin_ctz((unsigned)N);    // N is a power of two: shift / mask instead of a division sequence
    if ((long)(blockIdx.x >> nsh) * kGemmVec >= B) return;   // whole tile past the active list (uniform)
    constexpr int A_UNITS = K * 8;           // 16-byte units of the entries tile per stage
    constexpr int A_PER_THREAD = (A_UNITS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *ldsA = reinterpret_cast<f32x4 *>(smem);
    f32x4 *ldsB = ldsA + A_UNITS;

    const int n = blockIdx.x & (N - 1);
    const long b0 = (long)(blockIdx.x >> nsh) * kGemmVec;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;

    const float *Bn = Bm + (long)n * K * Dp;
    const int xstride = (MODE == MODE_STAGE0) ? Dp : D;
    const bool x_vec = (MODE == MODE_STAGE0) || (((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(xin) & 15) == 0));

    // staging assignment: unit f -> (row = f / 8, c = f % 8 -> kb = c / 4, g = c % 4).
    // Every staging load is UNCONDITIONAL (row and k indices are clamped into range instead of
    // guarded): a guarded load costs a branch and a full vmcnt(0) drain each.  Clamped loads fetch
    // data that is never used: rows past the batch are dropped in the epilogue, the k-block past Dp
    // of an odd tail is skipped by the MFMA loop.
    const float *oldrow[2] = {Bm, Bm};
    long brow[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const long row = b0 + ((tid + 256 * s) >> 3);
        brow[s] = row < B ? row : B - 1;
        if (MODE == MODE_STAGE0) oldrow[s] = Bm + ((long)n * K + idx_in[brow[s] * N + n]) * Dp;
    }
    // the x rows of the logits pass are unpadded: vector loads only when D is already a multiple of 16
    const bool fast = (MODE == MODE_STAGE0) || (x_vec && D == Dp && !xh);
    const _Float16 *xin_h = reinterpret_cast<const _Float16 *>(xin);
    const bool fast_h = (MODE != MODE_STAGE0) && xh && D == Dp && ((reinterpret_cast<uintptr_t>(xin) & 7) == 0);

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 stA[A_PER_THREAD], stB[2], stO[2];
    const int nkb = Dp / 16;
    const int nsteps = (nkb + 1) / 2;

    auto load_stage = [&](int step) {
        const int k0 = step * kGemmBK;
#pragma unroll
        for (int s = 0; s < A_PER_THREAD; ++s) {
            const int f = tid + 256 * s;
            int row = f >> 3;
            row = row < K ? row : K - 1;
            int k = k0 + 4 * (f & 7);
            k = k < Dp ? k : Dp - 4;
            stA[s] = *reinterpret_cast<const f32x4 *>(Bn + (long)row * Dp + k);
        }
        if (fast) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int k = k0 + 4 * ((tid + 256 * s) & 7);
                k = k < Dp ? k : Dp - 4;
                // raw loads only: the subtraction / scaling happens when the stage is stored to LDS, so
                // these loads stay in flight under the MFMAs of the current stage
                stB[s] = *reinterpret_cast<const f32x4 *>(xin + brow[s] * xstride + k);
                if (MODE == MODE_STAGE0) stO[s] = *reinterpret_cast<const f32x4 *>(oldrow[s] + k);
            }
        } else if (fast_h) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int k = k0 + 4 * ((tid + 256 * s) & 7);
                k = k < Dp ? k : Dp - 4;
                stB[s] = load_h4(xin_h + brow[s] * xstride + k);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = k0 + 4 * ((tid + 256 * s) & 7);
                const float *xr = xin + brow[s] * xstride;
                const _Float16 *xrh = xin_h + brow[s] * xstride;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ke = (k + e < xstride) ? k + e : 0;
                    const float val = xh ? (float)xrh[ke] : xr[ke];
                    v[e] = (k + e < xstride) ? val : 0.f;
                }
                stB[s] = v;
            }
        }
    };

    load_stage(0);
    for (int step = 0; step < nsteps; ++step) {
#pragma unroll
        for (int s = 0; s < A_PER_THREAD; ++s) {
            const int f = tid + 256 * s;
            if (f < A_UNITS) ldsA[lds_unit(K, f >> 3, (f & 7) >> 2, f & 3)] = stA[s];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int f = tid + 256 * s;
            const f32x4 v = (MODE == MODE_STAGE0) ? (stB[s] - stO[s]) : (MODE == MODE_XC ? stB[s] : stB[s] * lscale);
            ldsB[lds_unit(kGemmVec, f >> 3, (f & 7) >> 2, f & 3)] = v;
        }
        __syncthreads();
        if (step + 1 < nsteps) load_stage(step + 1);
        const int kbs = (2 * step + 1 < nkb) ? 2 : 1;
        for (int kb = 0; kb < kbs; ++kb) {
            const f32x4 bf = ldsB[lds_unit(kGemmVec, 16 * wave + r, kb, g)];
            f32x4 af[T];
#pragma unroll
            for (int t = 0; t < T; ++t) af[t] = ldsA[lds_unit(K, 16 * t + r, kb, g)];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int t = 0; t < T; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][i], bf[i], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: lane holds, for vector b0 + 16*wave + r, entries k = 16t + 4g + v
    const long b = b0 + 16 * wave + r;
    if (MODE == MODE_STAGE0) {
        if (b < B) {
            const float Rv = Rin[b * N + n];
            float *o = out + (b * N + n) * (long)K;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(Qin + (long)n * K + 16 * t + 4 * g);
                f32x4 s;
#pragma unroll
                for (int v = 0; v < 4; ++v) s[v] = (Rv + q[v]) + 2.0f * acc[t][v];
                *reinterpret_cast<f32x4 *>(o + 16 * t + 4 * g) = s;
            }
        }
    } else if (MODE == MODE_XC) {
        if (b < B) {
#pragma unroll
            for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4 *>(out + (b * N + n) * (long)K + 16 * t + 4 * g) = acc[t];
        }
    } else {
        float best = -INFINITY;
        int bk = 0;
        bool first = true;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + (long)n * K + 16 * t + 4 * g);
            f32x4 lv;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                lv[v] = acc[t][v] + bi[v];
                const int k = 16 * t + 4 * g + v;
                if (first || lv[v] > best) { best = lv[v]; bk = k; first = false; }
            }
            if (MODE == MODE_LOGITS_OUT && b < B)
                *reinterpret_cast<f32x4 *>(out + (b * N + n) * (long)K + 16 * t + 4 * g) = lv;
        }
        if (MODE == MODE_LOGITS || (MODE == MODE_LOGITS_OUT && idx_out != nullptr)) {
            // combine the 4 lanes of this vector: first maximum = greatest value, lowest k on ties
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float ov = __shfl_xor(best, m, 64);
                const int ok = __shfl_xor(bk, m, 64);
                const bool take = (ov > best) || (ov == best && ok < bk);
                best = take ? ov : best;
                bk = take ? ok : bk;
            }
            if (g == 0 && b < B) idx_out[b * N + n] = (uint8_t)bk;
        }
    }
}

// ---- 8/16-wave variant (K >= 32): the default GEMM -----------------------------------------
// Same tile and numerics as k_gemm, but the K entries are split over two groups of VGN waves: each
// wave owns 16 vectors x K/2 entries (half the accumulators and fragments: <= 128 VGPRs, 4 waves per
// SIMD).  A stage is ONE k-block (20 KB at K = 256) and there are two LDS buffers, so a wave stores
// stage s+1 right after issuing its MFMAs of stage s while the other waves of the SIMD are still
// computing; the loads of stage s+2 are issued after the barrier (one barrier per stage).
// Ablation (this kernel, dim 512): without the per-stage LDS stores the loop runs at 135 TFLOP/s,
// without the global loads 125, as is 111-115: LDS write bandwidth competing with the fragment
// reads is what the MFMA pipe waits for -- barriers are free, and writing the entries tile by
// LDS-DMA (global_load_lds) or from a 32-float single/double-buffered stage measured no better.
// one k-block stage: row-major [row][4 units], unit index xor-ed with bits 1-2 of the row:
// conflict-free for the fragment ds_read_b128 and for the staging ds_write_b128 (4 lanes per row),
// checked by brute force in tools/lds_conflicts.py
__device__ __forceinline__ int lds_unit1(int rows, int row, int g) { (void)rows; return row * 4 + (g ^ ((row >> 1) & 3)); }

template <int T, int MODE, int VGN>
__global__ void __launch_bounds__(128 * VGN, 4)
k_gemm8s(const float *__restrict__ Bm, const float *__restrict__ xin, const uint8_t *__restrict__ idx_in,
         float lscale, const float *__restrict__ bias, const float *__restrict__ Rin,
         const float *__restrict__ Qin, long B, int N, int D, int Dp, uint8_t *__restrict__ idx_out,
         float *__restrict__ out, int keep, const int *__restrict__ nact,
         const float *__restrict__ lscale_ptr /* overrides lscale when non-null */, int xh /* logits modes: x is fp16 */) {
    constexpr bool IS0 = (MODE == MODE_STAGE0) || (MODE == MODE_STAGE0_SEL);
    if (nact) B = *nact;
    if (lscale_ptr) lscale = *lscale_ptr;
    const int nsh = __builtin_ctz((unsigned)N);    // N is a power of two: shift / mask instead of a division sequence
    if ((long)(blockIdx.x >> nsh) * (16 * VGN) >= B) return;   // whole tile past the active list (uniform)
    static_assert(T >= 2 && T % 2 == 0, "k_gemm8s splits the entry tiles over two wave groups");
    constexpr int K = 16 * T;
    constexpr int TW = T / 2;
    constexpr int NT = 128 * VGN;             // VGN vector groups x 2 entry halves, one wave each
    constexpr int VEC = 16 * VGN;              // vectors per workgroup
    constexpr int A_UNITS = K * 4;                 // 16-byte units of one k-block of the entries tile
    constexpr int B_UNITS = VEC * 4;
    constexpr int A_PER_THREAD = (A_UNITS + NT - 1) / NT;
    constexpr int STAGE_UNITS = A_UNITS + B_UNITS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem);  // [2][STAGE_UNITS]

    const int n = blockIdx.x & (N - 1);
    const long b0 = (long)(blockIdx.x >> nsh) * VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vg = wave % VGN, eh = wave / VGN;
    const int r = lane & 15, g = lane >> 4;
    const float *Bn = Bm + (long)n * K * Dp;
    const int xstride = IS0 ? Dp : D;
    const bool x_vec = IS0 || (((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(xin) & 15) == 0));
    const bool fast = IS0 || (x_vec && D == Dp && !xh);
    const bool fast_h = !IS0 && xh && D == Dp && ((reinterpret_cast<uintptr_t>(xin) & 7) == 0);

    // staging: unit f -> (row = f / 4, g = f % 4); threads 0..255 also stage the vector tile
    const bool has_b = tid < B_UNITS;
    long browl = b0 + ((tid & (B_UNITS - 1)) >> 2);
    browl = browl < B ? browl : B - 1;
    const float *xbase = xin + b0 * xstride;
    const _Float16 *xbase_h = reinterpret_cast<const _Float16 *>(xin) + b0 * xstride;
    const uint32_t xoff = (uint32_t)((browl - b0) * xstride) + 4 * (tid & 3);
    uint32_t ooff = 0;
    if (IS0) ooff = (uint32_t)(((long)n * K + idx_in[browl * N + n]) * Dp) + 4 * (tid & 3);
    uint32_t aoff[A_PER_THREAD];
#pragma unroll
    for (int s = 0; s < A_PER_THREAD; ++s) {
        int row = (tid + NT * s) >> 2;
        row = row < K ? row : K - 1;
        aoff[s] = (uint32_t)(row * Dp) + 4 * (tid & 3);
    }

    f32x4 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 stA[A_PER_THREAD], stB = (f32x4){0.f, 0.f, 0.f, 0.f}, stO = stB;
    const int nkb = Dp / 16;

    auto load_stage = [&](int kb) {
        const uint32_t k = (uint32_t)(16 * kb);
#pragma unroll
        for (int s = 0; s < A_PER_THREAD; ++s) stA[s] = *reinterpret_cast<const f32x4 *>(Bn + (aoff[s] + k));
        if (has_b) {
            if (fast) {
                stB = *reinterpret_cast<const f32x4 *>(xbase + (xoff + k));
                if (IS0) stO = *reinterpret_cast<const f32x4 *>(Bm + (ooff + k));
            } else if (fast_h) {
                stB = load_h4(xbase_h + (xoff + k));
            } else {
                const int kk = 16 * kb + 4 * (tid & 3);
                const float *xr = xbase + (xoff - 4 * (tid & 3));
                const _Float16 *xrh = xbase_h + (xoff - 4 * (tid & 3));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ke = (kk + e < xstride) ? kk + e : 0;
                    const float val = xh ? (float)xrh[ke] : xr[ke];
                    stB[e] = (kk + e < xstride) ? val : 0.f;
                }
            }
        }
    };
    auto store_stage = [&](int buf) {
        f32x4 *sa = lds + (size_t)buf * STAGE_UNITS, *sb = sa + A_UNITS;
#pragma unroll
        for (int s = 0; s < A_PER_THREAD; ++s) {
            const int f = tid + NT * s;
            if (f < A_UNITS) sa[lds_unit1(K, f >> 2, f & 3)] = stA[s];
        }
        if (has_b) {
            const f32x4 v = IS0 ? (stB - stO) : (MODE == MODE_XC ? stB : stB * lscale);
            sb[lds_unit1(VEC, tid >> 2, tid & 3)] = v;
        }
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();
    if (nkb > 1) load_stage(1);
    // one k-block: fragments from buffer BUF (a compile-time constant: every LDS address of the loop body is a
    // loop-invariant register plus an immediate), MFMAs, then the next stage goes to the other buffer
    auto kstep = [&](int kb, auto BUF) {
        constexpr int buf = decltype(BUF)::value;
        const f32x4 *sa = lds + (size_t)buf * STAGE_UNITS, *sb = sa + A_UNITS;
        const f32x4 bf = sb[lds_unit1(VEC, 16 * vg + r, g)];
        f32x4 af[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) af[t] = sa[lds_unit1(K, 16 * (eh * TW + t) + r, g)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < TW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][i], bf[i], acc[t], 0, 0, 0);
        if (kb + 1 < nkb) store_stage(buf ^ 1);
        __syncthreads();
        if (kb + 2 < nkb) load_stage(kb + 2);
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        kstep(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < nkb) kstep(kb + 1, std::integral_constant<int, 1>{});
    }

    // epilogue: lane holds, for vector b0 + 16*vg + r, entries 16*(eh*TW+t) + 4g + v
    const long b = b0 + 16 * vg + r;
    if (MODE == MODE_STAGE0_SEL) {
        // Scores (R + Q) + 2X go to LDS, 32 vectors per round ([vector][K + 4] floats: the +4 keeps
        // the ds_write_b128 of 8 lanes = 8 rows on 8 different bank groups), then each wave selects
        // for 32 / (#waves) vectors with the same wave_select_fast as k_prune0.
        constexpr int RS = K + 4;
        constexpr int WAVES = 2 * VGN;
        constexpr int PER_WAVE = 32 / WAVES;
        constexpr int VPL = (K >= 64) ? K / 64 : 1;
        float *S_lds = reinterpret_cast<float *>(smem);
        u64 *scratch = reinterpret_cast<u64 *>(smem + 32 * RS * 4) + (size_t)wave * kSelectLdsU64;
        const float Rv = Rin[(b < B ? b : B - 1) * N + n];
        for (int round = 0; round < VEC / 32; ++round) {
            __syncthreads();   // the stage buffers / the previous round's scores are no longer read
            if ((vg >> 1) == round) {
                float *row = S_lds + ((vg & 1) * 16 + r) * RS;
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    const int k0 = 16 * (eh * TW + t) + 4 * g;
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(Qin + (long)n * K + k0);
                    f32x4 sv;
#pragma unroll
                    for (int v = 0; v < 4; ++v) sv[v] = (Rv + q[v]) + 2.0f * acc[t][v];
                    *reinterpret_cast<f32x4 *>(row + k0) = sv;
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int j = 0; j < PER_WAVE; ++j) {
                const int vloc = wave * PER_WAVE + j;
                const long bv = b0 + round * 32 + vloc;
                if (bv >= B) continue;   // wave-uniform
                const float *sr = S_lds + vloc * RS;
                float sv[VPL];
                int sp[VPL];
                if (K >= 256) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4 *>(sr + 4 * lane);
#pragma unroll
                    for (int i = 0; i < VPL; ++i) { sv[i] = t4[i & 3]; sp[i] = VPL * lane + i; }
                } else {
#pragma unroll
                    for (int i = 0; i < VPL; ++i) {
                        const int pos = VPL * lane + i;
                        const bool ok = pos < K;
                        sv[i] = ok ? sr[ok ? pos : 0] : INFINITY;
                        sp[i] = ok ? pos : kBigPos;
                    }
                }
                float ov;
                int op;
                wave_select_fast<VPL>(sv, sp, keep, K, scratch, ov, op);
                if (lane < keep) {
                    idx_out[(bv * N + n) * keep + lane] = (uint8_t)op;   // idx_out = tuples [B][N][keep]
                    out[(bv * N + n) * keep + lane] = ov;                 // out = scores [B][N][keep]
                }
            }
        }
    } else if (MODE == MODE_STAGE0) {
        if (b < B) {
            const float Rv = Rin[b * N + n];
            float *o = out + (b * N + n) * (long)K;
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const int k0 = 16 * (eh * TW + t) + 4 * g;
                const f32x4 q = *reinterpret_cast<const f32x4 *>(Qin + (long)n * K + k0);
                f32x4 sv;
#pragma unroll
                for (int v = 0; v < 4; ++v) sv[v] = (Rv + q[v]) + 2.0f * acc[t][v];
                *reinterpret_cast<f32x4 *>(o + k0) = sv;
            }
        }
    } else if (MODE == MODE_XC) {
        if (b < B) {
#pragma unroll
            for (int t = 0; t < TW; ++t)
                *reinterpret_cast<f32x4 *>(out + (b * N + n) * (long)K + 16 * (eh * TW + t) + 4 * g) = acc[t];
        }
    } else {
        float best = -INFINITY;
        int bk = 0;
        bool first = true;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int k0 = 16 * (eh * TW + t) + 4 * g;
            const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + (long)n * K + k0);
            f32x4 lv;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                lv[v] = acc[t][v] + bi[v];
                if (first || lv[v] > best) { best = lv[v]; bk = k0 + v; first = false; }
            }
            if (MODE == MODE_LOGITS_OUT && b < B) *reinterpret_cast<f32x4 *>(out + (b * N + n) * (long)K + k0) = lv;
        }
        if (MODE == MODE_LOGITS || (MODE == MODE_LOGITS_OUT && idx_out != nullptr)) {
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float ov = __shfl_xor(best, m, 64);
                const int ok = __shfl_xor(bk, m, 64);
                const bool take = (ov > best) || (ov == best && ok < bk);
                best = take ? ov : best;
                bk = take ? ok : bk;
            }
            float *cv = reinterpret_cast<float *>(smem);
            int *ck = reinterpret_cast<int *>(smem) + VEC;
            __syncthreads();   // every wave is done reading the last stage
            if (eh == 1 && g == 0) { cv[16 * vg + r] = best; ck[16 * vg + r] = bk; }
            __syncthreads();
            if (eh == 0 && g == 0 && b < B) {
                const float ov = cv[16 * vg + r];
                const int ok = ck[16 * vg + r];
                idx_out[b * N + n] = (uint8_t)((ov > best) ? ok : bk);
            }
        }
    }
}

// -------------------------------------------------------------------- prune0
// First sort-and-truncate (quantization.py:470-503 at L = 1): one wave per (b, n)
// keeps the `keep` smallest of S0[b][n][0..K).  keep == 1 happens only for N == 1,
// where the kept entry IS the new index (:468-469).
template <int K>
__global__ void k_prune0(const float *__restrict__ S0, long BN, int keep, uint8_t *__restrict__ tup_out,
                         float *__restrict__ S_out, uint8_t *__restrict__ idx_final, const int *__restrict__ nact,
                         int N) {
    if (nact) BN = (long)*nact * N;
    constexpr int VPL = (K >= 64) ? K / 64 : 1;
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= BN) return;
    const int lane = lane_id();
    float v[VPL];
    int p[VPL];
    const float *s = S0 + w * K;
    if (K >= 256) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(s + 4 * lane);
#pragma unroll
        for (int i = 0; i < VPL; ++i) { v[i] = t[i & 3]; p[i] = VPL * lane + i; }
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int pos = VPL * lane + i;
            const bool ok = pos < K;
            v[i] = ok ? s[ok ? pos : 0] : INFINITY;
            p[i] = ok ? pos : kBigPos;
        }
    }
    float ov;
    int op;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *scratch = reinterpret_cast<u64 *>(smem) + (size_t)(threadIdx.x >> 6) * kSelectLdsU64;
    wave_select_fast<VPL>(v, p, keep, K, scratch, ov, op);
    if (lane < keep) {
        if (idx_final != nullptr) {
            idx_final[w] = (uint8_t)op;
        } else {
            tup_out[w * keep + lane] = (uint8_t)op;
            S_out[w * keep + lane] = ov;
        }
    }
}

// ---------------------------------------------------------------------- pair
// One combine + sort-and-truncate step (quantization.py:504-547 then :470-503):
// one wave per (vector b, output group go).  Input groups e = 2*go, o = 2*go + 1
// each hold KI candidates: a tuple of L codebook entries (codebooks e*L .. e*L+L-1)
// and a score.  delta(candidate) is rebuilt from the tuple as the reference built
// it: leaves c - old (:436-439) summed pairwise up the combine tree (:538-541).
//   S'[a*KI + b] = ((Se[a] + So[b]) - E) + 2 * dot16(delta_e[a], delta_o[b])
// The `keep` smallest (value, position) survive; their tuples are concatenated.
// MFMA rows = even-group candidates a, columns = odd-group candidates b.
// delta of one candidate from its L preloaded codebook pieces: leaves c - old (:436-439) summed
// pairwise up the combine tree (:538-541).  `old` pieces come from the wave's LDS window.
template <int L>
struct DeltaBuilder {
    template <int LL>
    static __device__ __forceinline__ f32x4 build(const f32x4 *raw /*[L]*/, const float *oldwin, int win, int j0,
                                                  int off /* float offset of this lane's piece in the window */) {
        if constexpr (LL == 1) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(oldwin + j0 * win + off);
            return raw[j0] - o;
        } else {
            const f32x4 lo = build<LL / 2>(raw, oldwin, win, j0, off);
            const f32x4 hi = build<LL / 2>(raw, oldwin, win, j0 + LL / 2, off);
            return lo + hi;
        }
    }
};

#ifndef MCQ_PAIR_UNR1
#define MCQ_PAIR_UNR1 2   // same, single-leaf 16x16 stage
#endif
#ifndef MCQ_PAIR_UNR
#define MCQ_PAIR_UNR 2   // k-blocks per software-pipeline batch x leaves (4: deeper prefetch, one wave per SIMD fewer)
#endif
// DEDUP (L == 4, KI == 32): a candidate of this stage is the sum of two candidates of the stage before the previous
// one's lists, delta_4 = delta_2P[a] + delta_2Q[b] (:538-541), and only 16 + 16 distinct delta_2 rows feed the 32
// candidates of a side.  The kernel gathers those 32 two-leaf rows per side (half the row fetches of rebuilding every
// candidate from its four leaves: this stage is bound by the L1 data rate), forms delta_2 = (c0 - o0) + (c1 - o1)
// once, parks the four delta_2 tiles in a wave-private LDS tile and builds each MFMA operand row as tile_P[a] +
// tile_Q[b] with per-lane row addresses -- the same additions in the same order as DeltaBuilder<4>.
//   pos_in  [B][Gin][KI][2]: (a, b) of every candidate, written by the previous stage (pos_out there)
//   tup_prev[B][2*Gin][16][2]: the two-leaf lists the previous stage combined
template <int L, int KI, bool XL = false, int ABL = 0, bool DEDUP = false>
__global__ void __launch_bounds__(256)
k_pair(const float *__restrict__ C, const uint8_t *__restrict__ idx, const float *__restrict__ E,
       const uint8_t *__restrict__ tup_in /*[B][Gin][KI][L]*/, const float *__restrict__ S_in /*[B][Gin][KI]*/,
       long B, int N, int K, int Dp, int Gout, int keep, int win /* floats of each old row staged per window */,
       uint8_t *__restrict__ tup_out /*[B][Gout][keep][2L]*/, float *__restrict__ S_out,
       uint8_t *__restrict__ idx_final /* may alias idx: a wave only rewrites its own vector, at the end */,
       const int *__restrict__ nact, uint8_t *__restrict__ pos_out /* nullable: [B][Gout][keep][2] */,
       const uint8_t *__restrict__ pos_in, const uint8_t *__restrict__ tup_prev) {
    static_assert(!DEDUP || (L == 4 && KI == 32 && ABL == 0 && !XL), "DEDUP is the 32x32 four-leaf stage");
    constexpr int TI = (KI + 15) / 16;
    if (nact) B = *nact;
    constexpr int VPL = TI * TI * 4;
    constexpr int M = KI * KI;
    constexpr bool SMALL = L <= 4;            // row offsets precomputed in registers
    constexpr int TW = (L + 3) / 4;           // packed tuple words per candidate (large L)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int wpb = blockDim.x >> 6;
    // all waves of a workgroup work on the SAME output group for consecutive vectors, and
    // workgroup id mod Gout picks the group: workgroups land on XCD (id mod 8), so each XCD's
    // L2 only ever sees the codebooks of the groups congruent to it (2L*K rows instead of N*K).
    const int go = (int)(blockIdx.x & (unsigned)(Gout - 1));       // Gout is a power of two
    const long b = (long)(blockIdx.x >> __builtin_ctz((unsigned)Gout)) * wpb + wave;
    if (b >= B) return;   // no workgroup-wide barrier below: every LDS region is private to its wave
    const int r = lane & 15, g = lane >> 4;
    const int Gin = 2 * Gout;
    const int ge = 2 * go, gd = 2 * go + 1;
    const int n0 = ge * L;   // codebooks n0 .. n0 + 2L - 1 belong to this pair of groups

    u64 *scratch = reinterpret_cast<u64 *>(smem) + (size_t)wave * kSelectLdsU64;
    float *oldwin = reinterpret_cast<float *>(smem) + (size_t)wpb * kSelectLdsU64 * 2 + (size_t)wave * 2 * L * win;
    // XL: operands change lane order through a wave-private LDS tile (ds_write_b128 + ds_read_b128:
    // 13.4 + 4.3 LDS cycles per KB) instead of four ds_bpermute (4 x 6.2) -- tools/micro/lds_write_rates.hip
    constexpr int XP_UNITS = 2 * TI * 64;
    f32x4 *xpose = reinterpret_cast<f32x4 *>(smem + (size_t)wpb * (kSelectLdsU64 * 8 + (size_t)2 * L * win * 4)) +
                   (size_t)wave * XP_UNITS;

    // Operand rows are LOADED in a coalescing-friendly lane order -- lane 4*rs + ps reads the
    // ps-th float4 of the k-block of candidate row rs, so each quad of lanes covers 64 contiguous
    // bytes (16 cache accesses per wave-load instead of 64) -- and moved to the MFMA operand
    // order (lane 16*g + r holds row r, float4 g) with four ds_bpermute per float4.
    // (quads 8..15 hold their four parts rotated by two so that the 32 lanes of a bpermute
    // half-wave pull from 32 distinct LDS-crossbar banks)
    const int rs = lane >> 2, ps = (XL || DEDUP) ? (lane & 3) : ((lane & 3) ^ ((lane >> 5) << 1));
    const int perm_addr = (4 * r + (g ^ ((r >> 3) << 1))) << 2;   // byte address of this lane's source lane
    // LDS tile units (16 B): row-major 4 per row, the quad index xor (row / 4) % 4 keeps the 16 lanes of a
    // ds_read_b128 group (fixed g) on 16 distinct bank quads
    const int xw = 4 * rs + (ps ^ ((rs >> 2) & 3)), xr = 4 * r + (g ^ ((r >> 2) & 3));
    const uint8_t *te = tup_in + ((b * Gin + ge) * KI) * (long)L;
    const uint8_t *to = tup_in + ((b * Gin + gd) * KI) * (long)L;
    // Everything the epilogue needs from memory is requested NOW (E, the candidates' scores, and for
    // L <= 4 the candidates' tuples, lane j holding tuple j of both groups), so that after the MFMA
    // loop no global round trip is left: the selected tuples then come from registers by ds_bpermute.
    const float Eb = E[b];
    float se_pre[TI][4], so_pre[TI];
    {
        const float *Se = S_in + (b * Gin + ge) * (long)KI;
        const float *So = S_in + (b * Gin + gd) * (long)KI;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int bcol = 16 * ti + r;
            so_pre[ti] = So[bcol < KI ? bcol : 0];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int a = 16 * ti + 4 * g + v;
                se_pre[ti][v] = Se[a < KI ? a : 0];
            }
        }
    }
    constexpr bool TUP_IN_REGS = (L <= 4);
    uint32_t tup_e = 0, tup_o = 0;   // lane j < KI: the L bytes of tuple j of the even / odd group
    if constexpr (TUP_IN_REGS) {
        const int cj = lane < KI ? lane : 0;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            tup_e |= (uint32_t)te[cj * L + j] << (8 * j);
            tup_o |= (uint32_t)to[cj * L + j] << (8 * j);
        }
    }
    bool validA[TI], validB[TI];
    // side 0 = even group (MFMA A / rows), side 1 = odd group (MFMA B / columns)
    uint32_t coff[2][TI][SMALL ? L : 1];
    uint32_t tw[2][TI][SMALL ? 1 : TW];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
        validA[ti] = validB[ti] = (16 * ti + r) < KI;
        const int cand = 16 * ti + rs;
        const int cc = cand < KI ? cand : 0;
        if constexpr (SMALL) {
#pragma unroll
            for (int j = 0; j < L; ++j) {
                coff[0][ti][j] = 4u * (uint32_t)(((n0 + j) * K + te[cc * L + j]) * Dp + 4 * ps);
                coff[1][ti][j] = 4u * (uint32_t)(((n0 + L + j) * K + to[cc * L + j]) * Dp + 4 * ps);
            }
        } else {
#pragma unroll
            for (int w = 0; w < TW; ++w) {   // tuples of L >= 8 bytes are 4-byte aligned
                tw[0][ti][w] = *reinterpret_cast<const uint32_t *>(te + cc * L + 4 * w);
                tw[1][ti][w] = *reinterpret_cast<const uint32_t *>(to + cc * L + 4 * w);
            }
        }
    }
    auto row_off = [&](int side, int ti, int j) -> uint32_t {   // BYTE offset; side, ti, j compile-time at every call
        if constexpr (SMALL) {
            return coff[side][ti][j];
        } else {
            const uint32_t e = (tw[side][ti][j >> 2] >> (8 * (j & 3))) & 0xffu;
            return 4u * (uint32_t)(((n0 + side * L + j) * K + (int)e) * Dp + 4 * ps);
        }
    };
    const char *Cb = reinterpret_cast<const char *>(C);   // byte offsets (< 2^32) from the uniform base
    const f32x4 abl_const = {Eb, Eb + 1.f, Eb + 2.f, Eb + 3.f};
#define MCQ_PAIR_GATHER(expr) ((ABL == 4 || ABL == 7) ? abl_const : *reinterpret_cast<const f32x4 *>(expr))
    auto to_mfma_order = [&](f32x4 v, int slot) {
        f32x4 o;
        if constexpr (ABL == 3 || ABL == 7) {
            return v;
        } else if constexpr (XL) {
            f32x4 *t = xpose + slot * 64;
            t[xw] = v;
            asm volatile("" ::: "memory");   // same-wave LDS accesses execute in order; keep the compiler from swapping them
            o = t[xr];
            asm volatile("" ::: "memory");
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                o[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_addr, __float_as_int(v[c])));
        }
        return o;
    };

    // stage floats [w0, w0 + wlen) of the 2L old rows into this wave's LDS window
    auto stage_old = [&](int w0, int wlen) {
        const int per_row = wlen / 4;
        constexpr int JB = (2 * L < 8) ? 2 * L : 8;
        for (int j0 = 0; j0 < 2 * L; j0 += JB) {
            uint32_t rb[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u)
                rb[u] = (uint32_t)(((n0 + j0 + u) * K + idx[b * N + n0 + j0 + u]) * Dp + w0);
            for (int q = lane; q < per_row; q += 128) {
                const int q2 = q + 64;
                const bool v2 = q2 < per_row;
                const int q2c = v2 ? q2 : q;
                f32x4 t0[JB], t1[JB];
#pragma unroll
                for (int u = 0; u < JB; ++u) {
                    t0[u] = *reinterpret_cast<const f32x4 *>(C + rb[u] + 4 * q);
                    t1[u] = *reinterpret_cast<const f32x4 *>(C + rb[u] + 4 * q2c);
                }
#pragma unroll
                for (int u = 0; u < JB; ++u) {
                    float *dst = oldwin + (size_t)(j0 + u) * win;
                    *reinterpret_cast<f32x4 *>(dst + 4 * q) = t0[u];
                    if (v2) *reinterpret_cast<f32x4 *>(dst + 4 * q2) = t1[u];
                }
            }
        }
    };

    f32x4 acc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mfma_block = [&](const f32x4 (&da)[TI], const f32x4 (&db)[TI]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < TI; ++tj) {
                    if constexpr (ABL == 2 || ABL == 7) acc[ti][tj][i] += da[ti][i] + db[tj][i];
                    else acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[ti][i], db[tj][i], acc[ti][tj], 0, 0, 0);
                }
    };
    auto finish_operand = [&](f32x4 d, bool valid, int slot) {
        d = to_mfma_order(d, slot);
        if (KI < 16 && !valid) d = (f32x4){0.f, 0.f, 0.f, 0.f};  // padded rows of an 8-candidate group
        return d;
    };

    constexpr bool PIPE = (L * TI) <= 2;
    constexpr int UNR = PIPE ? (L * TI == 1 ? MCQ_PAIR_UNR1 : MCQ_PAIR_UNR / (L * TI)) : 1;

    for (int w0 = 0; w0 < (ABL == 1 ? 0 : Dp); w0 += win) {
        const int wlen = (Dp - w0 < win) ? (Dp - w0) : win;
        if (w0 > 0) wave_lds_fence();      // every read of the previous window has been issued
        if constexpr (ABL != 5) stage_old(w0, wlen);     // 5: old rows left as found (timing only)
        wave_lds_fence();
        const int kb_lo = w0 / 16, nkb = wlen / 16;   // k-blocks of this window
        const float *oldp = oldwin + 4 * ps;          // this lane's piece within a k-block of the window

        if constexpr (DEDUP) {
            typedef __attribute__((address_space(3))) volatile f32x4 lds_vf4;
            lds_vf4 *vt = (lds_vf4 *)xpose;                      // four tiles [side][P|Q] of 16 rows x 4 units
            const int Gp = 2 * Gin;                              // groups of the lists two stages back
            uint32_t doff[2][2][2];                              // [side][P|Q][leaf]: row rs of that two-leaf list
            int ua[2][TI], ub[2][TI];                            // LDS units of this lane's operand rows
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const int gs = sd == 0 ? ge : gd;
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    const uint8_t *tp = tup_prev + ((b * Gp + 2 * gs + pq) * 16 + rs) * 2L;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        doff[sd][pq][j] = 4u * (uint32_t)(((n0 + sd * L + pq * 2 + j) * K + tp[j]) * Dp + 4 * ps);
                }
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const uint8_t *pp = pos_in + ((b * Gin + gs) * KI + 16 * ti + r) * 2L;
                    const int a = pp[0] & 15, q = pp[1] & 15;
                    ua[sd][ti] = (sd * 2 + 0) * 64 + 4 * a + (g ^ ((a >> 2) & 3));
                    ub[sd][ti] = (sd * 2 + 1) * 64 + 4 * q + (g ^ ((q >> 2) & 3));
                }
            }
            const int xwu = 4 * rs + ((lane & 3) ^ ((rs >> 2) & 3));   // (this path loads with ps = lane & 3: see below)
            auto dgather = [&](f32x4 (&c)[2][2][2], int kbi) {
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            c[sd][pq][j] = *reinterpret_cast<const f32x4 *>(Cb + 64 * (size_t)(kb_lo + kbi) + (size_t)doff[sd][pq][j]);
            };
            auto dcompute = [&](const f32x4 (&c)[2][2][2], int kbi) {
                f32x4 d2[2][2];
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq) {
                        const float *o = oldp + (size_t)(sd * L + pq * 2) * win + 16 * kbi;
                        const f32x4 o0 = *reinterpret_cast<const f32x4 *>(o), o1 = *reinterpret_cast<const f32x4 *>(o + win);
                        d2[sd][pq] = (c[sd][pq][0] - o0) + (c[sd][pq][1] - o1);
                    }
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq) vt[(sd * 2 + pq) * 64 + xwu] = d2[sd][pq];
                f32x4 da[TI], db[TI];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const f32x4 pa = vt[ua[0][ti]], qa = vt[ub[0][ti]];
                    const f32x4 pb = vt[ua[1][ti]], qb = vt[ub[1][ti]];
                    da[ti] = pa + qa;
                    db[ti] = pb + qb;
                }
                mfma_block(da, db);
            };
            f32x4 c0[2][2][2], c1[2][2][2];
            dgather(c0, 0);
            for (int kbi = 0; kbi < nkb; kbi += 2) {
                const int k1 = (kbi + 1 < nkb) ? kbi + 1 : kbi, k2 = (kbi + 2 < nkb) ? kbi + 2 : kbi;   // clamped re-reads
                dgather(c1, k1);
                __builtin_amdgcn_sched_barrier(0);
                dcompute(c0, kbi);
                __builtin_amdgcn_sched_barrier(0);
                dgather(c0, k2);
                __builtin_amdgcn_sched_barrier(0);
                if (kbi + 1 < nkb) dcompute(c1, kbi + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (PIPE) {
            // Two-deep software pipeline over batches of UNR k-blocks: the gathers of batch c+1 are
            // issued (pinned by sched_barrier) before the arithmetic of batch c.  Inside the
            // steady-state loop every load is unconditional, so the compiler's vmcnt bookkeeping is
            // exact and the waits are counted, not drains.
            auto load_batch = [&](f32x4 (&ra)[UNR][TI][L], f32x4 (&rb)[UNR][TI][L], int kb0) {
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                        for (int j = 0; j < L; ++j) {
                            ra[u][ti][j] = MCQ_PAIR_GATHER(Cb + (row_off(0, ti, j) + (uint32_t)(64 * (kb_lo + kb0 + u))));
                            rb[u][ti][j] = MCQ_PAIR_GATHER(Cb + (row_off(1, ti, j) + (uint32_t)(64 * (kb_lo + kb0 + u))));
                        }
            };
            auto compute_batch = [&](const f32x4 (&ra)[UNR][TI][L], const f32x4 (&rb)[UNR][TI][L], int kb0) {
                if constexpr (XL && ABL == 0) {
                    // LDS-tile transposition, batched: every `old` read and subtraction of the batch first, then the
                    // tile writes / reads as VOLATILE accesses (kept in program order by the compiler, executed in
                    // order by the LDS: a slot is reused by the next k-block without any wait), so the only
                    // lgkmcnt waits left are the counted ones in front of the MFMAs.
                    f32x4 d[UNR][2][TI], o[UNR][2][TI];
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int ti = 0; ti < TI; ++ti) {
                            d[u][0][ti] = DeltaBuilder<L>::template build<L>(ra[u][ti], oldp, win, 0, 16 * (kb0 + u));
                            d[u][1][ti] = DeltaBuilder<L>::template build<L>(rb[u][ti], oldp + L * win, win, 0, 16 * (kb0 + u));
                        }
                    typedef __attribute__((address_space(3))) volatile f32x4 lds_vf4;   // keep the accesses DS (not FLAT)
                    lds_vf4 *vt = (lds_vf4 *)xpose;
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                            for (int ti = 0; ti < TI; ++ti) vt[(sd * TI + ti) * 64 + xw] = d[u][sd][ti];
#pragma unroll
                        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                            for (int ti = 0; ti < TI; ++ti) {
                                f32x4 t = vt[(sd * TI + ti) * 64 + xr];
                                if (KI < 16 && !(sd == 0 ? validA[ti] : validB[ti])) t = (f32x4){0.f, 0.f, 0.f, 0.f};
                                o[u][sd][ti] = t;
                            }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) mfma_block(o[u][0], o[u][1]);
                    return;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    f32x4 da[TI], db[TI];
#pragma unroll
                    for (int ti = 0; ti < TI; ++ti) {
                        da[ti] = finish_operand(DeltaBuilder<L>::template build<L>(ra[u][ti], oldp, win, 0, 16 * (kb0 + u)), validA[ti], ti);
                        db[ti] = finish_operand(DeltaBuilder<L>::template build<L>(rb[u][ti], oldp + L * win, win, 0, 16 * (kb0 + u)), validB[ti], TI + ti);
                    }
                    mfma_block(da, db);
                }
            };
            const int nb = nkb / UNR;   // whole batches; the nkb % UNR leftover k-blocks follow one by one
            f32x4 r0a[UNR][TI][L], r0b[UNR][TI][L], r1a[UNR][TI][L], r1b[UNR][TI][L];
            int c = 0;
            if (nb > 0) {
                load_batch(r0a, r0b, 0);
                while (c + 2 < nb) {
                    load_batch(r1a, r1b, (c + 1) * UNR);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(r0a, r0b, c * UNR);
                    __builtin_amdgcn_sched_barrier(0);
                    load_batch(r0a, r0b, (c + 2) * UNR);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(r1a, r1b, (c + 1) * UNR);
                    __builtin_amdgcn_sched_barrier(0);
                    c += 2;
                }
                if (nb - c == 2) {
                    load_batch(r1a, r1b, (c + 1) * UNR);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(r0a, r0b, c * UNR);
                    __builtin_amdgcn_sched_barrier(0);
                    compute_batch(r1a, r1b, (c + 1) * UNR);
                } else {
                    compute_batch(r0a, r0b, c * UNR);
                }
            }
            for (int kbi = nb * UNR; kbi < nkb; ++kbi) {
                f32x4 ta[TI][L], tb[TI][L];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        ta[ti][j] = MCQ_PAIR_GATHER(Cb + 64 * (size_t)(kb_lo + kbi) + (size_t)row_off(0, ti, j));
                        tb[ti][j] = MCQ_PAIR_GATHER(Cb + 64 * (size_t)(kb_lo + kbi) + (size_t)row_off(1, ti, j));
                    }
                f32x4 da[TI], db[TI];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    da[ti] = finish_operand(DeltaBuilder<L>::template build<L>(ta[ti], oldp, win, 0, 16 * kbi), validA[ti], ti);
                    db[ti] = finish_operand(DeltaBuilder<L>::template build<L>(tb[ti], oldp + L * win, win, 0, 16 * kbi), validB[ti], TI + ti);
                }
                mfma_block(da, db);
            }
        } else if constexpr (L * TI <= 8) {
            // Medium shapes: all 2*TI*L gathers of a k-block are in flight together (the other waves of
            // the SIMD cover their latency; deeper per-wave prefetch measured slower here: the stage is
            // bound by L1 tag throughput, not by latency).
            f32x4 ra[TI][L], rb[TI][L];
            auto load_all = [&](int kbi) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        ra[ti][j] = MCQ_PAIR_GATHER(Cb + 64 * (size_t)(kb_lo + kbi) + (size_t)row_off(0, ti, j));
                        rb[ti][j] = MCQ_PAIR_GATHER(Cb + 64 * (size_t)(kb_lo + kbi) + (size_t)row_off(1, ti, j));
                    }
            };
            for (int kbi = 0; kbi < nkb; ++kbi) {
                load_all(kbi);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 da[TI], db[TI];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    da[ti] = finish_operand(DeltaBuilder<L>::template build<L>(ra[ti], oldp, win, 0, 16 * kbi), validA[ti], ti);
                    db[ti] = finish_operand(DeltaBuilder<L>::template build<L>(rb[ti], oldp + L * win, win, 0, 16 * kbi), validB[ti], TI + ti);
                }
                mfma_block(da, db);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // Heavy shapes (L*TI > 8): the 2*TI operand tiles of a k-block stream through two
            // rotating L-leaf register buffers (tile t+1 is gathered while tile t is summed), and
            // tile 0 of the NEXT k-block is gathered before this k-block's MFMAs.
            auto load_tile = [&](f32x4 (&buf)[L], int side, int ti, int kbi) {
#pragma unroll
                for (int j = 0; j < L; ++j)
                    buf[j] = MCQ_PAIR_GATHER(Cb + 64 * (size_t)(kb_lo + kbi) + (size_t)row_off(side, ti, j));
            };
            f32x4 bufA[L], bufB[L];
            load_tile(bufA, 0, 0, 0);
            for (int kbi = 0; kbi < nkb; ++kbi) {
                f32x4 da[TI], db[TI];
                const int nxt = (kbi + 1 < nkb) ? kbi + 1 : kbi;   // clamped: the last prefetch is a harmless re-read
#pragma unroll
                for (int t = 0; t < 2 * TI; ++t) {
                    // tiles in the order A0, B0, A1, B1, ...; even t lives in bufA, odd t in bufB
                    const int side = t & 1, ti = t >> 1;
                    if (t + 1 < 2 * TI) {
                        if (((t + 1) & 1) == 0) load_tile(bufA, 0, (t + 1) >> 1, kbi);
                        else load_tile(bufB, 1, (t + 1) >> 1, kbi);
                    } else {
                        load_tile(bufA, 0, 0, nxt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (side == 0) da[ti] = finish_operand(DeltaBuilder<L>::template build<L>(bufA, oldp, win, 0, 16 * kbi), validA[ti], ti);
                    else db[ti] = finish_operand(DeltaBuilder<L>::template build<L>(bufB, oldp + L * win, win, 0, 16 * kbi), validB[ti], TI + ti);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma_block(da, db);
            }
        }
    }

    // scores: lane holds rows a = 16*ti + 4*g + v, column bcol = 16*tj + r
    float sv[VPL];
    int sp[VPL];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TI; ++tj) {
            const int bcol = 16 * tj + r;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int a = 16 * ti + 4 * g + v;
                const bool ok = (a < KI) && (bcol < KI);
                const float val = ((se_pre[ti][v] + so_pre[tj]) - Eb) + 2.0f * acc[ti][tj][v];
                const int slot = (ti * TI + tj) * 4 + v;
                sv[slot] = ok ? val : INFINITY;
                sp[slot] = ok ? a * KI + bcol : kBigPos;
            }
        }
    float ov;
    int op;
    if constexpr (ABL == 6) {        // 6: no final selection (timing only)
        ov = sv[0];
        op = sp[0] & (M - 1);
    } else {
        wave_select_fast<VPL>(sv, sp, keep, M, scratch, ov, op);
    }
    // lane j < keep owns output candidate j = (a, bb) in the input lists
    const int a = (lane < keep ? op : 0) / KI, bb = (lane < keep ? op : 0) % KI;
    uint32_t we = 0, wo = 0;
    if constexpr (TUP_IN_REGS) {   // all lanes take part in the cross-lane reads
        we = (uint32_t)__builtin_amdgcn_ds_bpermute(a << 2, (int)tup_e);
        wo = (uint32_t)__builtin_amdgcn_ds_bpermute(bb << 2, (int)tup_o);
    }
    if (lane < keep) {
        // last step (one group, keep == 1): the tuple is the new index vector (:468-469)
        uint8_t *o = (idx_final != nullptr) ? idx_final + b * N
                                            : tup_out + ((b * Gout + go) * (long)keep + lane) * (2 * L);
        if constexpr (TUP_IN_REGS) {
#pragma unroll
            for (int j = 0; j < L; ++j) { o[j] = (uint8_t)(we >> (8 * j)); o[L + j] = (uint8_t)(wo >> (8 * j)); }
        } else {
            for (int j = 0; j < L; ++j) { o[j] = te[a * L + j]; o[L + j] = to[bb * L + j]; }
        }
        if (idx_final == nullptr) S_out[(b * Gout + go) * (long)keep + lane] = ov;
        if (pos_out != nullptr) {       // which two list entries this candidate combines (read by a DEDUP stage)
            uint8_t *po = pos_out + ((b * Gout + go) * (long)keep + lane) * 2;
            po[0] = (uint8_t)a;
            po[1] = (uint8_t)bb;
        }
    }
}

// ------------------------------------------------- pair combine, 8-candidate lists
// The ladders of 16-entry codebooks (the trainer's first phase) combine lists of 8 candidates: 8 x 8 scores fill
// a quarter of a 16 x 16 MFMA tile.  This kernel packs TWO output groups into one wave: MFMA rows 0-7 / columns 0-7
// belong to output group 2*gp, rows 8-15 / columns 8-15 to group 2*gp + 1 (the off-diagonal blocks are computed and
// dropped), so a stage needs half the waves and half the MFMAs of k_pair<L, 8>.  Same arithmetic, operand order and
// selection as k_pair: results are bit-identical.  `old` rows are read straight from L2 (each is shared by the 8
// rows of its half: two 64-byte segments per load), so there is no staging prologue.  L in {1, 2}.
template <int L>
__global__ void __launch_bounds__(64)
k_pair8(const float *__restrict__ C, const uint8_t *__restrict__ idx, const float *__restrict__ E,
        const uint8_t *__restrict__ tup_in /*[B][Gin][8][L]*/, const float *__restrict__ S_in /*[B][Gin][8]*/, long B,
        int N, int K, int Dp, int Gout, int keep, uint8_t *__restrict__ tup_out /*[B][Gout][keep][2L]*/,
        float *__restrict__ S_out, uint8_t *__restrict__ idx_final, const int *__restrict__ nact) {
    constexpr int KI = 8;
    if (nact) B = *nact;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *scratch = reinterpret_cast<u64 *>(smem);
    const int GP = (Gout + 1) / 2;
    const int gp = (int)(blockIdx.x & (unsigned)(GP - 1));         // GP is a power of two
    const long b = (long)(blockIdx.x >> __builtin_ctz((unsigned)GP));
    if (b >= B) return;
    const int lane = lane_id();
    const int r = lane & 15, g = lane >> 4;
    const int Gin = 2 * Gout;
    // load layout: lane 4*rs + ps reads float4 ps of the k-block of packed row rs (half rs / 8, candidate rs % 8)
    const int rs = lane >> 2, ps = (lane & 3) ^ ((lane >> 5) << 1);
    const int perm_addr = (4 * r + (g ^ ((r >> 3) << 1))) << 2;
    const int hl = rs >> 3, cl = rs & 7;                          // this lane's half / candidate in the load layout
    const int go_l = (2 * gp + hl < Gout) ? 2 * gp + hl : 2 * gp;   // an absent second half re-reads the first (dropped later)
    const int n0_l = 2 * go_l * L;
    const uint8_t *te_l = tup_in + ((b * Gin + 2 * go_l) * KI) * (long)L;
    const uint8_t *to_l = tup_in + ((b * Gin + 2 * go_l + 1) * KI) * (long)L;
    uint32_t coff[2][L], ooff[2][L];     // byte offsets of this lane's piece: candidate leaves / old leaves, per side
#pragma unroll
    for (int j = 0; j < L; ++j) {
        coff[0][j] = 4u * (uint32_t)(((n0_l + j) * K + te_l[cl * L + j]) * Dp + 4 * ps);
        coff[1][j] = 4u * (uint32_t)(((n0_l + L + j) * K + to_l[cl * L + j]) * Dp + 4 * ps);
        ooff[0][j] = 4u * (uint32_t)(((n0_l + j) * K + idx[b * N + n0_l + j]) * Dp + 4 * ps);
        ooff[1][j] = 4u * (uint32_t)(((n0_l + L + j) * K + idx[b * N + n0_l + L + j]) * Dp + 4 * ps);
    }
    // epilogue inputs, requested now: E, and the list scores of this lane's MFMA rows / column
    const float Eb = E[b];
    const int hrow = g >> 1, hcol = r >> 3;                     // half of this lane's rows 4g+v / of its column r
    const int go_row = (2 * gp + hrow < Gout) ? 2 * gp + hrow : 2 * gp;
    const int go_col = (2 * gp + hcol < Gout) ? 2 * gp + hcol : 2 * gp;
    float se_pre[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) se_pre[v] = S_in[(b * Gin + 2 * go_row) * (long)KI + ((4 * g + v) & 7)];
    const float so_pre = S_in[(b * Gin + 2 * go_col + 1) * (long)KI + (r & 7)];

    const char *Cb = reinterpret_cast<const char *>(C);
    auto gather = [&](f32x4 (&c)[2][L], f32x4 (&o)[2][L], int kb) {
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int j = 0; j < L; ++j) {
                c[sd][j] = *reinterpret_cast<const f32x4 *>(Cb + (coff[sd][j] + (uint32_t)(64 * kb)));
                o[sd][j] = *reinterpret_cast<const f32x4 *>(Cb + (ooff[sd][j] + (uint32_t)(64 * kb)));
            }
    };
    auto operand = [&](const f32x4 (&c)[L], const f32x4 (&o)[L]) {
        f32x4 d = c[0] - o[0];                              // leaves c - old (:436-439) ...
        if constexpr (L == 2) d = d + (c[1] - o[1]);        // ... summed up the combine tree (:538-541)
        f32x4 t;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_addr, __float_as_int(d[q])));
        return t;
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nkb = Dp / 16;
    f32x4 c0[2][L], o0[2][L], c1[2][L], o1[2][L];
    gather(c0, o0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {
        const int k1 = (kb + 1 < nkb) ? kb + 1 : kb, k2 = (kb + 2 < nkb) ? kb + 2 : kb;   // clamped: harmless re-reads
        gather(c1, o1, k1);
        __builtin_amdgcn_sched_barrier(0);
        {
            const f32x4 da = operand(c0[0], o0[0]), db = operand(c0[1], o0[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da[i], db[i], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        gather(c0, o0, k2);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) {   // uniform
            const f32x4 da = operand(c1[0], o1[0]), db = operand(c1[1], o1[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da[i], db[i], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // scores of the two diagonal 8 x 8 blocks; one selection per half
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int go = 2 * gp + h;
        if (go >= Gout) break;   // uniform
        float sv[4];
        int sp[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int a = 4 * g + v;
            const bool ok = (hrow == h) && (hcol == h);
            sv[v] = ok ? ((se_pre[v] + so_pre) - Eb) + 2.0f * acc[v] : INFINITY;
            sp[v] = ok ? (a & 7) * KI + (r & 7) : kBigPos;
        }
        float ov;
        int op;
        wave_select_fast<4>(sv, sp, keep, KI * KI, scratch, ov, op);
        if (lane < keep) {
            const int a = op / KI, bb = op % KI;
            const uint8_t *te = tup_in + ((b * Gin + 2 * go) * KI) * (long)L;
            const uint8_t *to = tup_in + ((b * Gin + 2 * go + 1) * KI) * (long)L;
            uint8_t *o = (idx_final != nullptr) ? idx_final + b * N : tup_out + ((b * Gout + go) * (long)keep + lane) * (2 * L);
#pragma unroll
            for (int j = 0; j < L; ++j) { o[j] = te[a * L + j]; o[L + j] = to[bb * L + j]; }
            if (idx_final == nullptr) S_out[(b * Gout + go) * (long)keep + lane] = ov;
        }
    }
}

// ------------------------------------------------------- fixed-point skipping
// _refine_indexes is a deterministic map F of (x, indexes): once F(idx) == idx every later pass
// returns idx again, so such a vector can leave the active list without changing any result.
// One thread per active slot: converged (or last pass) -> the indexes go to their original row of
// `final_idx`; otherwise the slot is re-packed (order irrelevant: vectors are independent) for the
// next pass.  `cnt_next` was zeroed by the host (memset node ahead of the launch).
__global__ void k_compact(const uint8_t *__restrict__ idx_old, const uint8_t *__restrict__ idx_new,
                          const int *__restrict__ map_cur, const int *__restrict__ nact, long B, int N, int last,
                          uint8_t *__restrict__ final_idx, uint8_t *__restrict__ idx_packed, int *__restrict__ map_next,
                          int *__restrict__ cnt_next) {
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (nact) B = *nact;
    if (s >= B) return;
    const int orig = map_cur ? map_cur[s] : (int)s;
    bool changed = false;
    if (!last)
        for (int n = 0; n < N; ++n) changed = changed || (idx_old[s * N + n] != idx_new[s * N + n]);
    if (changed) {
        const int slot = atomicAdd(cnt_next, 1);
        for (int n = 0; n < N; ++n) idx_packed[(long)slot * N + n] = idx_new[s * N + n];
        map_next[slot] = orig;
    } else {
        for (int n = 0; n < N; ++n) final_idx[(long)orig * N + n] = idx_new[s * N + n];
    }
}

// -------------------------------------------------------------------- output
// encode tail (quantization.py:266-275): uint8 with nibble packing when K == 16
// (low nibble = even codebook, :269), or int64 indexes.
__global__ void k_finalize(const uint8_t *__restrict__ idx, long B, int N, int pack, uint8_t *__restrict__ out_u8,
                           int64_t *__restrict__ out_i64) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (out_i64 != nullptr) {
        if (i < B * N) out_i64[i] = idx[i];
    } else {
        const int per = N / pack;
        if (i < B * per) {
            if (pack == 1) out_u8[i] = idx[i];
            else out_u8[i] = (uint8_t)(idx[2 * i] + 16 * idx[2 * i + 1]);
        }
    }
}

// int64 indexes supplied by the caller -> the uint8 working copy of the search (values clamped to K-1)
__global__ void k_import_indexes(const int64_t *__restrict__ in, long n, int K, uint8_t *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        long v = in[i];
        v = v < 0 ? 0 : (v > K - 1 ? K - 1 : v);
        out[i] = (uint8_t)v;
    }
}

// -------------------------------------------------------------------- decode
// out[b][:] = sum_n C[n][index(b, n)][:D], n ascending (quantization.py:131-148).
// One wave per vector; codes are uint8 or int64, optionally packed r digits per code
// (least significant first, _maybe_separate_indexes :551-573).
template <typename CodeT>
__global__ void k_decode(const CodeT *__restrict__ codes, int per_row, long B, const float *__restrict__ C, int N,
                         int K, int D, int Dp, float *__restrict__ out) {
    const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = lane_id();
    const int rep = N / per_row;
    const CodeT *cb = codes + b * per_row;
    float *ob = out + b * D;
    const bool vec_ok = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int q = lane; q < Dp / 4; q += 64) {
        f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < N; ++n) {
            long code = (long)cb[n / rep];
            int digit = n % rep;
            for (int d = 0; d < digit; ++d) code /= K;
            const int k = (int)(code % K) & (K - 1);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(C + ((long)n * K + k) * Dp + 4 * q);
            t = (n == 0) ? c : t + c;
        }
        if (vec_ok && 4 * q + 3 < D) {
            *reinterpret_cast<f32x4 *>(ob + 4 * q) = t;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (4 * q + c < D) ob[4 * q + c] = t[c];
        }
    }
}

// XCD-sliced decode: workgroup id mod 8 (= the XCD it lands on) picks one eighth of the feature axis, so
// every XCD's L2 only ever holds its own slice of the codebooks (N*K*Dp/8 floats: 0.5 MB at dim 512 / 8
// codebooks instead of 4 MB, which is the whole L2).  A wave covers 64 / LPV vectors, LPV lanes x float4
// per vector slice; rows are added n ascending in chunks of CH gathers in flight.  Unpacked codes only.
template <typename CodeT, int CH, int LPV>
__global__ void __launch_bounds__(256)
k_decode_sliced(const CodeT *__restrict__ codes, long B, const float *__restrict__ C, int N, int K, int D, int Dp,
                float *__restrict__ out) {
    constexpr int VPW = 64 / LPV;
    const int slice = blockIdx.x & 7;
    const long vb = blockIdx.x >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = lane / LPV, q = lane % LPV;
    const long b = (vb * 4 + wave) * VPW + v;
    const int off = slice * (LPV * 4) + 4 * q;
    if (b >= B || off >= Dp) return;
    const CodeT *cb = codes + b * N;
    const float *Cq = C + off;
    f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += CH) {
        f32x4 rows[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int n = (n0 + j < N) ? n0 + j : N - 1;
            const int k = (int)cb[n] & (K - 1);
            rows[j] = *reinterpret_cast<const f32x4 *>(Cq + ((long)n * K + k) * Dp);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (n0 + j < N) t = (n0 + j == 0) ? rows[j] : t + rows[j];
    }
    float *ob = out + b * D + off;
    if (((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && off + 3 < D) {
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(ob));   // streamed: keep the L2 for the codebooks
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (off + c < D) ob[c] = t[c];
    }
}

// LDS-resident decode for very large batches: the feature axis is cut into slices of 16 floats (64 B of every
// codebook row); a persistent workgroup copies ITS slice of all N*K rows into LDS once (N*K*64 B, 128 KB at
// 8 x 256) and then serves a contiguous range of vectors from LDS: the gathers no longer cross the L2->L1
// fabric (measured ceiling ~17 TB/s chip-wide, i.e. 2.1 TB/s of output at 8 codebooks), only the codes
// come in and 64 B per (vector, slice) go out.  Slices 4x..4x+3 sit on XCD x so that both halves of an
// output cache line pass through one L2.  Same sums in the same order as k_decode.
template <typename CodeT>
__global__ void __launch_bounds__(1024)
k_decode_lds(const CodeT *__restrict__ codes, long B, const float *__restrict__ C, int N, int K, int D, int Dp,
             int groups /* workgroups per slice */, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *rows = reinterpret_cast<f32x4 *>(smem);            // [N*K][4]
    const int ns = Dp / 16;
    // workgroup -> (slice, group): consecutive ids go round the XCDs; XCD x takes slices congruent to
    // 4x..4x+3 (mod 32), each slice gets `groups` workgroups
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;          // within: 0 .. ns*groups/8 - 1
    const int per_xcd = (ns + 7) / 8;                                  // slices per XCD
    const int slice = xcd * per_xcd + within % per_xcd;
    const int grp = within / per_xcd;
    if (slice >= ns) return;
    const int tid = threadIdx.x;
    const int nrows = N * K;
    for (int u = tid; u < nrows * 4; u += blockDim.x)
        rows[u] = *reinterpret_cast<const f32x4 *>(C + (long)(u >> 2) * Dp + slice * 16 + 4 * (u & 3));
    __syncthreads();
    const long per = (B + groups - 1) / groups;
    const long b_lo = grp * per, b_hi = (b_lo + per < B) ? b_lo + per : B;
    const int q = tid & 3;
    const int off = slice * 16 + 4 * q;
    const bool vec_store = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && off + 3 < D;
    for (long b = b_lo + (tid >> 2); b < b_hi; b += blockDim.x >> 2) {
        const CodeT *cb = codes + b * N;
        f32x4 t = rows[((int)cb[0] & (K - 1)) * 4 + q];
        for (int n = 1; n < N; ++n) t = t + rows[(n * K + ((int)cb[n] & (K - 1))) * 4 + q];
        float *ob = out + b * D + off;
        if (vec_store) {
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(ob));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (off + c < D) ob[c] = t[c];
        }
    }
}

// Fast path for unpacked uint8 codes and the common small shapes: all NN x J row pieces of a
// vector are requested before the first add (16 gathers in flight per lane at dim 512 / 8 codebooks).
template <int NN, int J>
__global__ void k_decode_reg(const uint8_t *__restrict__ codes, long B, const float *__restrict__ C, int K, int D,
                             int Dp, float *__restrict__ out) {
    const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = lane_id();
    const uint8_t *cb = codes + b * NN;
    float *ob = out + b * D;
    const bool vec_ok = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const int nq = Dp / 4;
    f32x4 rows[NN][J];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const float *o = C + ((long)n * K + (cb[n] & (K - 1))) * Dp;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int q = lane + 64 * j;
            rows[n][j] = *reinterpret_cast<const f32x4 *>(o + 4 * (q < nq ? q : 0));
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int q = lane + 64 * j;
        f32x4 t = rows[0][j];
#pragma unroll
        for (int n = 1; n < NN; ++n) t = t + rows[n][j];
        if (q < nq) {
            if (vec_ok && 4 * q + 3 < D) {
                *reinterpret_cast<f32x4 *>(ob + 4 * q) = t;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * q + c < D) ob[4 * q + c] = t[c];
            }
        }
    }
}

// ----------------------------------------------------------- decode backward
// One wave per (codebook row (n, k), 64-feature chunk): scan the n-th index column of all B vectors
// 64 at a time (ballot) and add grad_out[b][chunk] for every match in ascending b -- a fixed
// summation order, so training is bit-reproducible (torch's index_add_ on the device uses atomics).
// Matching rows are fetched eight at a time so the loads overlap; the adds stay in order.
// Generalised to per-(vector, codebook) gradients: the value added for vector b into row (n, k) is
// gout[b * gsb + n * gsn + d] (decode: gsb = D, gsn = 0); idx[b * idx_stride + n]; negative indexes match no row.
template <typename IdxT>   // int64 indexes, or uint8 codes (8x less index traffic: the scan is what bounds this kernel)
__global__ void k_decode_backward(const float *__restrict__ gout, const IdxT *__restrict__ idx, long B, int N, int K,
                                  int D, int chunks, float *__restrict__ gC, long gsb, long gsn, int idx_stride,
                                  const float *__restrict__ sa = nullptr, const float *__restrict__ sb = nullptr, float sc = 1.0f,
                                  const float *__restrict__ dotw = nullptr, float *__restrict__ dot_part = nullptr) {
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= (long)N * K * chunks) return;
    const int lane = lane_id();
    const long row = w / chunks;
    const int d = (int)(w % chunks) * 64 + lane;
    const bool dok = d < D;
    const int dc = dok ? d : 0;
    const int n = (int)(row / K), k = (int)(row % K);
    float acc = 0.f;
    // index loads in flight per scan step: with 1-byte codes several steps' worth are fetched together (with
    // 8-byte indexes that floods the L1 with uncoalesced lines and measured slower)
    constexpr int SC = sizeof(IdxT) == 1 ? 8 : 1;
    for (long bs = 0; bs < B; bs += 64 * SC) {
        long iv[SC];
#pragma unroll
        for (int q = 0; q < SC; ++q) {
            const long b = bs + 64 * q + lane;
            iv[q] = (long)idx[(b < B ? b : B - 1) * idx_stride + n];
        }
#pragma unroll
        for (int q = 0; q < SC; ++q) {
            const long b0 = bs + 64 * q;
            const bool hit = (b0 + lane < B) && (iv[q] == (long)k);
            unsigned long long m = __ballot(hit);
            while (m) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    v[u] = 0.f;
                    if (m) {   // wave-uniform
                        const int l = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        v[u] = gout[(b0 + l) * gsb + n * gsn + dc];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = acc + v[u];   // x + 0 is exact: padding slots change nothing
            }
        }
    }
    // optional epilogue (trainer): the stored rows are scaled by f = sa[0] * sb[0] * sc, and the wave's share of
    // <sum, dotw> (the UNscaled sums against another [N*K][D] table) goes to dot_part[w] for a fixed-order reduction
    const float f = (sa ? *sa : 1.0f) * (sb ? *sb : 1.0f) * sc;
    if (dok) gC[row * D + d] = (sa || sb || sc != 1.0f) ? acc * f : acc;
    if (dot_part != nullptr) {
        float pd = dok ? acc * dotw[row * D + d] : 0.f;
        pd = wave_sum_butterfly(pd);
        if (lane == 0) dot_part[w] = pd;
    }
}

}  // namespace mcq
