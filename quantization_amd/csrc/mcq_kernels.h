// mcq_kernels.h -- gfx950 (CDNA4, wave64) kernels of the multi-codebook quantizer: selection, the GEMMs of the
// index search (logits + argmax, x.C, Gram matrix), the encode tail, decode and its backward.  The refinement pass
// itself (table form) is in mcq_tf_kernels.h.
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
//   * everything else is a single IEEE fp32 operation in the order the oracle writes it;
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
    // (bitwise on purpose: with || and && the compiler builds the short circuit out of exec masks and branches, some
    // fourteen instructions per call against seven)
    const bool take = (ov < v) | ((ov == v) & (op < p));
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
            const bool gt = (v[i] > pv) | ((v[i] == pv) & (p[i] > pp));
            const bool lt = (v[i] < bv) | ((v[i] == bv) & (p[i] < bp));
            if (gt & lt) { bv = v[i]; bp = p[i]; }
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
#ifndef MCQ_SEL_WIN
#define MCQ_SEL_WIN 4      // window of the selection's bound (wave_kth_lane_key); 0 = the exact key
#endif
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

// The key with exactly `target` smaller keys among the 64 per-lane keys `k` (unique; kKeyMax = no key): a quickselect whose
// candidate set is a wave-uniform 64-bit mask handled by scalar instructions.  One round = first candidate as the pivot,
// one ballot, one popcount; written as a do-while with both successor masks formed unconditionally (the while-with-break
// form compiled to 22 instructions and four branches per round, this one to about 14 and one).  kKeyMax when there are
// fewer than target + 1 keys.
// `win` > 0: ANY key with target .. target + win smaller ones will do (the caller only needs an upper bound of the target-th
// key that not many more keys lie below): the loop stops at the first pivot that lands in the window -- about two rounds of seven
// earlier for win = 4.
__device__ __forceinline__ u64 wave_kth_lane_key(u64 k, int target, int win = 0) {
    u64 cm = __ballot(k != kKeyMax);
    if (__popcll(cm) <= target) return kKeyMax;
    u64 kp;
    int rr;
    do {                          // every round removes at least the pivot's lane from the candidates: <= 64 rounds
#ifdef MCQ_DEBUG_SELECT
        // debug builds (hipcc -DMCQ_DEBUG_SELECT; __graft_entry__.build(debug_select=True)): the invariant below, checked.  An
        // empty candidate mask means two equal keys reached the selection -- trap instead of spinning on ctz(0)
        if (cm == 0) __builtin_trap();
#endif
        const int pl = __builtin_ctzll(cm);      // (cm != 0 here; __ffsll's zero case cost two scalar instructions per round)
        kp = readlane_u64(k, pl);
        const u64 ltm = __ballot(k < kp), gtm = __ballot(k > kp);      // (keys are unique: the lanes above the pivot, from a second
        rr = __popcll(ltm);                                            // vector compare instead of two more scalar mask operations)
        cm &= (rr > target) ? ltm : gtm;
    } while ((unsigned)(rr - target) > (unsigned)win);
    // INVARIANT the loop's exit rests on: the keys are UNIQUE (every caller packs the candidate's position into the low word), so
    // the key with exactly `target` smaller ones exists among the candidates and is reached before they run out.  A guard on
    // cm != 0 (equal keys would empty the mask first) was tried on the advisor's suggestion: two more scalar instructions in a loop
    // of fourteen, run seven times per selection, fourteen selections per vector and pass -- k_tf_stage0 268 -> 279 us, k_tf_pair0
    // 174 -> 179 us on one box; the invariant is asserted where the keys are built instead (tests/test_gpu_parity.py::
    // test_wave_selection_paths drives ties through every path: equal SCORES give distinct keys).
    return kp;
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
        // (a bound is all T0 has to be: any lane minimum with cnt - 1 .. cnt + 3 smaller ones.  Up to cnt + 4 lanes then hold
        // survivors -- more than 64 of them only if those lanes hold nearly all their keys below T0, in which case the exact
        // cnt-th minimum is taken after all: at most cnt * VPL <= 64 survive that)
        u64 T0 = wave_kth_lane_key(lmin, target, MCQ_SEL_WIN);
        // (T0 == kKeyMax: fewer than cnt lanes hold a candidate.  Every candidate then survives -- at most (cnt - 1) * VPL < 64 of
        // them -- and the non-candidates, whose key IS kKeyMax, must not: the bound becomes kKeyMax - 1 (a candidate's key is
        // below it: its low word is a position), and the retry below, which would return kKeyMax again, is skipped)
        const bool few = (T0 == kKeyMax);
        if (few) T0 = kKeyMax - 1;
        int base;
        for (int attempt = 0;; ++attempt) {
            base = 0;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const bool sel = key[i] <= T0;
                const u64 m = __ballot(sel);
                const int dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                if (sel && dst < 64) ldsA[dst] = key[i];
                base += __popcll(m);
            }
            if (base <= 64 || attempt > 0 || few) break;
            T0 = wave_kth_lane_key(lmin, target);
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
    if (VPL > 1 && cnt <= 64) {
        // Two-level variant with up to two survivors per lane (32 of 256 with four keys per lane): T0 as above bounds
        // the answer, at most cnt lanes hold keys <= T0, so at most cnt * VPL keys survive; lane l ranks the survivors
        // l and l + 64 against all of them.  With more keys per lane (32 of 1,024: sixteen) cnt * VPL exceeds the 128
        // slots, but the survivors rarely do (about 46 on the bench workload): they are counted first, and only a
        // count above 128 falls through to the general quickselect below (sixteen ballots per round).
        u64 lmin = key[0];
#pragma unroll
        for (int i = 1; i < VPL; ++i) lmin = key[i] < lmin ? key[i] : lmin;
        const u64 T0 = wave_kth_lane_key(lmin, target);
        int nsurv = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) nsurv += __popcll(__ballot(key[i] <= T0));
        if (nsurv <= 128) {
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
        const int c8 = (c0 + 7) & ~7;
        int ra = 0;
        if (c0 <= 64) {
            // the usual case (about 41 survivors for 32 of 256): nobody holds a second survivor, half the compares
            for (int j = 0; j < c8; j += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) ra += (ldsS[j + u] < ka) ? 1 : 0;
            }
        } else {
            const u64 kb = ldsS[lane + 64 < c0 ? lane + 64 : c0];      // (slot c0 holds a sentinel)
            int rb = 0;
            for (int j = 0; j < c8; j += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const u64 o = ldsS[j + u];
                    ra += (o < ka) ? 1 : 0;
                    rb += (o < kb) ? 1 : 0;
                }
            }
            if (lane + 64 < c0 && rb < cnt) ldsO[rb] = kb;
        }
        if (lane < c0 && ra < cnt) ldsO[ra] = ka;
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

// Test hook: one wave selects the `cnt` smallest of M = 64 * VPL scores (position = index) with wave_select_fast; lane j
// writes the j-th.  Lets the tests drive every path of the selection with adversarial inputs (ties, all survivors in a few
// lanes, more survivors than the two-per-lane path holds).
template <int VPL>
__global__ void __launch_bounds__(64)
k_test_select(const float *__restrict__ scores, int cnt, float *__restrict__ out_v, int *__restrict__ out_p) {
    __shared__ u64 scratch[kSelectLdsU64];
    const float *sc = scores + (size_t)blockIdx.x * 64 * VPL;
    float v[VPL];
    int p[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        p[i] = VPL * lane_id() + i;
        v[i] = sc[p[i]];
    }
    float ov;
    int op;
    wave_select_fast<VPL>(v, p, cnt, 64 * VPL, scratch, ov, op);
    if (lane_id() < cnt) {
        out_v[(size_t)blockIdx.x * 64 + lane_id()] = ov;
        out_p[(size_t)blockIdx.x * 64 + lane_id()] = op;
    }
}

// ------------------------------------------------------------------- prepare
// One wave per row: dst[row][0..Dp) = scale * src[row][0..D) zero padded; Q[row] = sumsq64.
// (get_centers(), quantization.py:77-79; all_centers_sumsq, :411)
__global__ void k_prepare_rows(const float *__restrict__ src, float scale, int apply_scale, long rows, int D,
                               int Dp, float *__restrict__ dst, float *__restrict__ Q,
                               const float *__restrict__ scale_ptr /* overrides `scale` when non-null */,
                               float *__restrict__ scales_out /* optional: scale_ptr[0..1] is copied here */,
                               const float *__restrict__ raw_cs = nullptr, const float *__restrict__ raw_ls = nullptr,
                               float speed = 0.f, float *__restrict__ scales_out2 = nullptr) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = lane_id();
    if (scale_ptr) {
        scale = *scale_ptr;
        if (scales_out && row == 0 && lane < 2) scales_out[lane] = scale_ptr[lane];
    }
    if (raw_cs) {
        // the scale PARAMETERS themselves (mcq_prepare_params): exp(speed * centers_scale) is formed here by every wave (the
        // same expf as mcq_scales_exp: same bits) and row 0 leaves both factors for the kernels that follow
        scale = expf(*raw_cs * speed);
        if (row == 0 && lane == 0) {
            const float ls = expf(*raw_ls * speed);
            if (scales_out) { scales_out[0] = scale; scales_out[1] = ls; }
            if (scales_out2) { scales_out2[0] = scale; scales_out2[1] = ls; }
        }
    }
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

// (the inner-product tables of the path -- logits, x.C, the Gram matrix -- are formed by mcq_fix_kernels.h)

// ------------------------------------------------------- fixed-point skipping
// _refine_indexes is a deterministic map F of (x, indexes): once F(idx) == idx every later pass
// returns idx again, so such a vector can leave the active list without changing any result.
// One thread per active slot: converged (or last pass) -> the indexes go to their original row of
// `final_idx`; otherwise the slot is re-packed (order irrelevant: vectors are independent) for the
// next pass.  `cnt_next` was zeroed by the host (memset node ahead of the launch).
template <typename CT>
__global__ void k_compact(const CT *__restrict__ idx_old, const CT *__restrict__ idx_new,
                          const int *__restrict__ map_cur, const int *__restrict__ nact, long B, int N, int last,
                          CT *__restrict__ final_idx, CT *__restrict__ idx_packed, int *__restrict__ map_next,
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
template <typename CT>
__global__ void k_finalize(const CT *__restrict__ idx, long B, int N, int pack, uint8_t *__restrict__ out_u8,
                           int64_t *__restrict__ out_i64, uint8_t *__restrict__ codes_also) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (out_i64 != nullptr) {
        if (i < B * N) {
            out_i64[i] = idx[i];
            if (codes_also) codes_also[i] = (uint8_t)idx[i];
        }
    } else {
        const int per = N / pack;
        if (i < B * per) {
            if (pack == 1) out_u8[i] = (uint8_t)idx[i];
            else out_u8[i] = (uint8_t)(idx[2 * i] + 16 * idx[2 * i + 1]);
        }
    }
}

// int64 indexes supplied by the caller -> the working copy of the search (one or two bytes each; values clamped to K-1)
template <typename CT>
__global__ void k_import_indexes(const int64_t *__restrict__ in, long n, int K, CT *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        long v = in[i];
        v = v < 0 ? 0 : (v > K - 1 ? K - 1 : v);
        out[i] = (CT)v;
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
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(ob));   // streamed: keep the L2 for the codebooks (plain stores measured slower here: 22.8 vs 18.0 us at dim 256 / 4 codebooks)
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
// The loop over a workgroup's vectors is latency bound (codes -> LDS addresses -> store), so the codes of UNR vectors
// are requested together, one trip ahead of their use: a trip then costs one global round trip for UNR vectors
// instead of one per vector (65,536 vectors at 8 x 256: 51.5 -> 36.9 us).  (32-byte rows, which would fit 16 x 256
// codebooks, measured slower than the sliced kernel: 332 vs 217 us at dim 1024.)
template <typename CodeT>
__global__ void __launch_bounds__(1024)
k_decode_lds(const CodeT *__restrict__ codes, long B, const float *__restrict__ C, int N, int K, int D, int Dp,
             int groups /* workgroups per slice */, float *__restrict__ out) {
    constexpr int W = 16, LPV = W / 4;                         // 64-byte rows, 4 lanes per (vector, slice)
    constexpr int UNR = 4;                                     // measured: 2 -> 46.7, 4 -> 36.9, 8 -> 42.0 us at 8 x 256, 65,536 vectors
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *rows = reinterpret_cast<f32x4 *>(smem);            // [N*K][LPV]
    const int ns = Dp / W;
    // workgroup -> (slice, group): consecutive ids go round the XCDs; XCD x takes a run of consecutive slices (both
    // halves of an output cache line pass through one L2), each slice gets `groups` workgroups
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int per_xcd = (ns + 7) / 8;                                  // slices per XCD
    const int slice = xcd * per_xcd + within % per_xcd;
    const int grp = within / per_xcd;
    if (slice >= ns) return;
    const int tid = threadIdx.x;
    const int nrows = N * K;
    for (int u = tid; u < nrows * LPV; u += blockDim.x)
        rows[u] = *reinterpret_cast<const f32x4 *>(C + (long)(u / LPV) * Dp + slice * W + 4 * (u % LPV));
    __syncthreads();
    const long per = (B + groups - 1) / groups;
    const long b_lo = grp * per, b_hi = (b_lo + per < B) ? b_lo + per : B;
    if (b_lo >= b_hi) return;
    const int q = tid % LPV;
    const int off = slice * W + 4 * q;
    const bool vec_store = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && off + 3 < D;
    const long stride = blockDim.x / LPV;
    // codes of one vector as two 64-bit words (N <= 16 bytes) when they are bytes and rows of N bytes are aligned
    const bool packed = sizeof(CodeT) == 1 && (N == 8 || N == 16) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
    auto fetch = [&](long b, unsigned long long (&w)[2]) {
        const long bc = b < b_hi ? b : b_hi - 1;
        const unsigned long long *p = reinterpret_cast<const unsigned long long *>(reinterpret_cast<const uint8_t *>(codes) + bc * N);
        w[0] = p[0];
        w[1] = (N == 16) ? p[1] : 0ull;
    };
    auto emit = [&](long b, const f32x4 &t) {
        if (b >= b_hi) return;
        float *ob = out + b * D + off;
        if (vec_store) {
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(ob));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (off + c < D) ob[c] = t[c];
        }
    };
    if (packed) {
        unsigned long long cur[UNR][2], nxt[UNR][2];
        long b = b_lo + tid / LPV;
#pragma unroll
        for (int u = 0; u < UNR; ++u) fetch(b + u * stride, cur[u]);
        for (; b < b_hi; b += UNR * stride) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) fetch(b + (UNR + u) * stride, nxt[u]);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                f32x4 t = rows[((int)(cur[u][0] & 0xffull) & (K - 1)) * LPV + q];
                for (int n = 1; n < N; ++n) {
                    const int code = (int)((cur[u][n >> 3] >> (8 * (n & 7))) & 0xffull) & (K - 1);
                    t = t + rows[(n * K + code) * LPV + q];
                }
                emit(b + u * stride, t);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; }
        }
        return;
    }
    for (long b = b_lo + tid / LPV; b < b_hi; b += stride) {
        const CodeT *cb = codes + b * N;
        f32x4 t = rows[((int)cb[0] & (K - 1)) * LPV + q];
        for (int n = 1; n < N; ++n) t = t + rows[(n * K + ((int)cb[n] & (K - 1))) * LPV + q];
        emit(b, t);
    }
}

// Block-staged LDS-resident decode for packed byte codes (N = 4, 8 or 16 per vector) whose row slices fit the LDS.
// What bounded its predecessor (a per-lane software pipeline: codes two trips ahead in registers) was not the gather-sums (23 us at dim 512 / 8 x 256 / 65,536 vectors with the stores removed) nor
// the write pattern (tools/micro/write_patterns.hip: the same 64-byte pieces written with PLAIN stores leave the chip at
// the rate of a contiguous fill, 21 us; nontemporal ones take 34 us) but the wait between them: on gfx9 loads and stores
// share one in-order counter (vmcnt), so waiting for the codes of a later trip also waits for every store issued before
// their load, and the compiler's register copies at the end of a trip made that a full drain per trip (s_waitcnt vmcnt(0)).
// Here no vector load is waited for inside the trips: the codes of a block of VB vectors (16 KB) go from global memory
// straight into LDS (global_load_lds_dwordx4, one 1 KB piece per wave, requested a whole block ahead) and the trips read
// them with ds_read; the one wait per block is s_waitcnt vmcnt(TRIPS) -- the piece was requested before the block's TRIPS
// stores, which stay in flight.  W = 4 * LPV floats per slice: 64-byte slices (LPV = 4), or 32-byte slices (LPV = 2) for
// 16 x 256 codebooks, whose 64-byte slices (256 KB) do not fit; XCD x owns a run of adjacent slices, so the pieces of an
// output cache line meet in one L2 and leave it as whole lines (plain stores).  Sums n ascending as in k_decode.
// A workgroup's last, partial block takes the guarded path (compiler-managed loads, per-vector clamps).
template <int N, int LPV>
__global__ void __launch_bounds__(1024)
k_decode_blk(const uint8_t *__restrict__ codes, long B, const float *__restrict__ C, int K, int D, int Dp,
             int groups /* workgroups per slice */, long per /* vectors per workgroup, a multiple of 256 */, float *__restrict__ out) {
    static_assert(N == 4 || N == 8 || N == 16, "packed codes: 4, 8 or 16 per vector");
    constexpr int W = 4 * LPV;
    constexpr int VPT = 1024 / LPV;                 // vectors per trip
    constexpr int VB = 1024 * 16 / N;               // vectors per block: 16 bytes of codes per thread
    constexpr int TRIPS = VB / VPT;
    static_assert(TRIPS >= 1 && TRIPS < 32, "");
    struct __attribute__((aligned(N))) cw_t { unsigned w[N / 4]; __device__ unsigned operator[](int i) const { return w[i]; } };      // the N code bytes of a vector
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *rows = reinterpret_cast<f32x4 *>(smem);                     // [N*K][LPV]
    char *cbuf = smem + (size_t)N * K * W * 4;                         // 2 x 16 KB of codes
    const int ns = Dp / W;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int per_xcd = (ns + 7) / 8;
    const int slice = xcd * per_xcd + within % per_xcd;
    const int grp = within / per_xcd;
    if (slice >= ns) return;
    const int tid = threadIdx.x;
    const long b_lo = grp * per, b_hi = (b_lo + per < B) ? b_lo + per : B;
    if (b_lo >= b_hi) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned voff = (unsigned)(tid & 63) * 16;
    // (m0 is written without being declared clobbered -- the compiler rejects it as a reserved register; nothing else in this
    // kernel uses m0: LDS instructions need no m0 on gfx9+)
    auto dma = [&](long b0, int buf) {              // this wave's 1 KB of the codes of block b0 -> cbuf[buf]
        const uint8_t *pg = codes + b0 * N + wave * 1024;
        const unsigned d = (unsigned)(size_t)cbuf + buf * 16384 + wave * 1024;
        asm volatile("s_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[p]" : : [vo] "v"(voff), [p] "s"(pg), [d] "s"(d) : "memory");
    };
    const bool first_full = b_lo + VB <= b_hi;
    if (first_full) dma(b_lo, 0);
    // the slice of every row, global -> LDS without passing through registers, all pieces of a wave in flight together
    // (a load / wait / ds_write loop costs one round trip per KB and thread: 8 of them at 8 x 256)
    for (int u0 = 0; u0 < N * K * LPV; u0 += 1024) {
        const int u = u0 + tid;
        if (u < N * K * LPV) {
            const unsigned go = (unsigned)(((long)(u / LPV) * Dp + slice * W + 4 * (u % LPV)) * 4);
            const unsigned d = (unsigned)(size_t)smem + (unsigned)(u0 + wave * 64) * 16;
            asm volatile("s_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[p]" : : [vo] "v"(go), [p] "s"(C), [d] "s"(d) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // (the first block's codes are in as well)
    const int q = tid % LPV, v0 = tid / LPV;
    const int off = slice * W + 4 * q;
    const bool lane_on = off < D;                    // (D % 4 == 0; the last slice may reach into the padding.  q = 0 is always inside)
    auto code_of = [&](const cw_t &w, int n) { return (int)((w[n >> 2] >> (8 * (n & 3))) & 0xffu) & (K - 1); };
    auto sum_rows = [&](const cw_t &w) {
        f32x4 t = rows[code_of(w, 0) * LPV + q];
#pragma unroll
        for (int n = 1; n < N; ++n) t = t + rows[(n * K + code_of(w, n)) * LPV + q];
        return t;
    };
    long b0 = b_lo;
    int buf = 0;
    for (; b0 + VB <= b_hi; b0 += VB, buf ^= 1) {
        // every wave has left the previous block (its reads of cbuf[buf ^ 1] are done) once all have arrived here
        if (b0 + 2 * VB <= b_hi) dma(b0 + VB, buf ^ 1);
        const char *cb = cbuf + buf * 16384;
        float *ob = out + (b0 + v0) * D + off;
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const cw_t w = *reinterpret_cast<const cw_t *>(cb + (size_t)(t * VPT + v0) * N);
            const f32x4 r = sum_rows(w);
            if (lane_on) *reinterpret_cast<f32x4 *>(ob + (long)t * VPT * D) = r;
        }
        // the next block's piece was requested before these TRIPS stores: they stay in flight
        if (b0 + 2 * VB <= b_hi) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" : : "n"(TRIPS) : "memory");
    }
    if (b0 < b_hi) {                                 // partial block
        for (long b = b0 + v0; b < b_hi; b += VPT) {
            const cw_t w = *reinterpret_cast<const cw_t *>(codes + b * N);
            const f32x4 r = sum_rows(w);
            if (lane_on) *reinterpret_cast<f32x4 *>(out + b * D + off) = r;
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
// CW = floats per lane: a wave covers 64 * CW features of its row, so the index column is scanned D / (64 CW) times per
// row instead of D / 64 (CW = 4 with float4 loads when rows are 16-byte aligned: 45.7 -> see DESIGN.md, trainer).
template <typename IdxT, int CW>   // int64 indexes, or uint8 codes (8x less index traffic: the scan is what bounds this kernel)
__global__ void k_decode_backward(const float *__restrict__ gout, const IdxT *__restrict__ idx, long B, int N, int K,
                                  int D, int chunks, float *__restrict__ gC, long gsb, long gsn, int idx_stride,
                                  const float *__restrict__ sa = nullptr, const float *__restrict__ sb = nullptr, float sc = 1.0f,
                                  const float *__restrict__ dotw = nullptr, float *__restrict__ dot_part = nullptr) {
    typedef float vecw __attribute__((ext_vector_type(CW)));
    // wave -> (row, feature chunk).  When the chunk count divides 8 a chunk belongs to 8 / chunks XCDs (workgroup id mod 8 =
    // the XCD): every codebook reads ALL vectors' gradients once, N times in total, and with the chunks spread over all XCDs
    // each L2 was asked to hold the whole gradient matrix (8 MB at 4,096 x 512: the kernel ran at the 3.5 TB/s of the
    // fabric behind the L2s, time proportional to the batch); an XCD that only ever sees its own columns keeps them
    // (1 MB at 8 chunks).  w = row * chunks + chunk stays the index of the wave's partial in dot_part.
    long row;
    int chunk;
    if (chunks <= 8 && (8 % chunks) == 0) {
        const int per = 8 / chunks, xcd = blockIdx.x & 7;
        chunk = xcd % chunks;
        row = ((long)(blockIdx.x >> 3) * per + xcd / chunks) * 4 + (threadIdx.x >> 6);
    } else {
        const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
        row = w0 / chunks;
        chunk = (int)(w0 % chunks);
    }
    if (row >= (long)N * K) return;
    const long w = row * chunks + chunk;
    const int lane = lane_id();
    const int d = (chunk * 64 + lane) * CW;       // CW > 1: D is a multiple of CW
    const bool dok = d < D;
    const int dc = dok ? d : 0;
    const int n = (int)(row / K), k = (int)(row % K);
    vecw acc;
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[c] = 0.f;
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
                vecw v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) v[u][c] = 0.f;
                    if (m) {   // wave-uniform
                        const int l = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        v[u] = *reinterpret_cast<const vecw *>(gout + (b0 + l) * gsb + n * gsn + dc);
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
    if (dok) *reinterpret_cast<vecw *>(gC + row * D + d) = (sa || sb || sc != 1.0f) ? acc * f : acc;
    if (dot_part != nullptr) {
        float pd = 0.f;
        if (dok) {
            const vecw wv = *reinterpret_cast<const vecw *>(dotw + row * D + d);
#pragma unroll
            for (int c = 0; c < CW; ++c) pd = pd + acc[c] * wv[c];
        }
        pd = wave_sum_butterfly(pd);
        if (lane == 0) dot_part[w] = pd;
    }
}

}  // namespace mcq
