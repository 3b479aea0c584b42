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
//   * selections keep the smallest candidates by (value, position), lowest position on ties, listed in ascending position.
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

// ---- selection, set form (round 6) ----------------------------------------------
// The sort-and-truncate of quantization.py:470-503 keeps the `cnt` smallest candidates; the reference needs the SET only
// (oracle/mcq_oracle.c::select_smallest), so the spec lists it in ascending POSITION and a wave hands its
// survivors over where they lie: no ranking of the survivors against each other, no sorted hand-over through LDS.
//   wave_select_set:  the cnt smallest of the wave's VPL*64 keys by (value, position), lowest position on equal values.
//   Result: every selected element sits in exactly one lane (`has`), with its index `dst` in the ascending-position list;
//   the caller stores it there.  Returns the number of selected elements (cnt, or the number of candidates if fewer).
// How: (a) one key per lane (64 candidates): an exact quickselect on the order-preserving 32-bit image of the score -- candidate
// sets are wave-uniform 64-bit masks on the scalar unit -- gives the threshold; equal scores across the boundary go to the lowest
// lanes; dst is a prefix count of the mask.  No LDS.  (b) several keys per lane: the cnt-th smallest (within a window) of the 64
// per-lane MINIMA bounds the answer; the survivors (about 1.5 cnt) are compacted, in position order (prefix counts over the
// slots' ballots), through 512 bytes of wave-private LDS into one per lane, and (a) runs on them.  (c) anything else (more
// than 64 survivors, positions that are not lane-major): an exact quickselect over all slots, then the same hand-over.
typedef unsigned long long u64;
constexpr u64 kKeyMax = ~0ull;
// fp16 ingestion (MCQ_ENCODE_X_FP16): x rows are _Float16 in HBM and widen to fp32 in the load path;
// every fp16 value is exactly representable, so the codes equal those of the widened input.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_h4(const void *p) {
    return __builtin_convertvector(*reinterpret_cast<const f16x4 *>(p), f32x4);
}
#ifndef MCQ_SEL_WIN
#define MCQ_SEL_WIN 4      // window of the bound on the lane minima (wave_kth_u32): any minimum with cnt - 1 .. cnt + 3 smaller ones
#endif
constexpr int kSelectLdsU64 = 128;  // per-wave LDS scratch of wave_select_set, in u64: 128 scores + 128 positions (one survivor per lane, two when they cluster)

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
// number of set bits of m below this lane, + base
__device__ __forceinline__ int mbcnt64(u64 m, int base = 0) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}

// Quickselect over one 32-bit key per lane.  `cm`: the lanes that hold a key (the others hold 0xffffffff and are never a pivot);
// more than `target` of them.  Returns a key kp with  n_lt(kp) <= target + win  and  n_le(kp) > target  (n_lt / n_le: keys below /
// not above kp), and the masks of the last round.  win = 0: THE key with at most `target` keys below it and more than `target`
// not above it (equal keys allowed: the caller settles them).  One round = first candidate as the pivot, two ballots, two
// popcounts; the key sought is never dropped (a pivot with too many keys below it keeps the lanes below it, any other pivot that
// is not accepted has n_le <= target and keeps the lanes above it), and every round drops at least the pivot's lane.
__device__ __forceinline__ uint32_t wave_kth_u32(uint32_t k, u64 cm, int target, int win, u64 &ltm, u64 &lem) {
    uint32_t kp;
    const int hi = target + win;
#ifdef MCQ_DEBUG_SELECT
    // debug builds (hipcc -DMCQ_DEBUG_SELECT; __graft_entry__.build(debug_select=True)): the same loop in C with the invariant
    // above CHECKED -- trap instead of spinning on ctz(0)
    for (;;) {
        if (cm == 0) __builtin_trap();
        const int pl = __builtin_ctzll(cm);
        kp = (uint32_t)__builtin_amdgcn_readlane((int)k, pl);
        ltm = __ballot(k < kp);
        lem = __ballot(k <= kp);
        if (__popcll(ltm) > hi) { cm &= ltm; continue; }
        if (__popcll(lem) > target) break;
        cm &= ~lem;
    }
#else
    // The loop as the ISA it should be: written in C (either as one loop condition over two flags or as the two branches above)
    // the compiler materialises both flags as 64-bit masks and selects between the successor masks -- 19 to 27 instructions a
    // round; here 9 when the pivot is too high (one vector compare), 13 otherwise.
    // (the candidate mask is COPIED into a register pair of the block's own: handed over as an in/out operand, a mask that is
    // __ballot(true) reaches the block as the exec register itself, and the loop would narrow the wave's exec mask)
    int t0;
    u64 cmw;
    asm("s_mov_b64 %[cm], %[cmin]\n\t"
        "1:\n\t"
        "s_ff1_i32_b64 %[t0], %[cm]\n\t"
        "v_readlane_b32 %[kp], %[k], %[t0]\n\t"
        "s_nop 1\n\t"
        "v_cmp_gt_u32_e64 %[lt], %[kp], %[k]\n\t"
        "s_bcnt1_i32_b64 %[t0], %[lt]\n\t"
        "s_cmp_gt_u32 %[t0], %[hi]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "v_cmp_ge_u32_e64 %[le], %[kp], %[k]\n\t"
        "s_bcnt1_i32_b64 %[t0], %[le]\n\t"
        "s_cmp_gt_u32 %[t0], %[tg]\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_andn2_b64 %[cm], %[cm], %[le]\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "s_and_b64 %[cm], %[cm], %[lt]\n\t"
        "s_branch 1b\n\t"
        "3:"
        : [kp] "=&s"(kp), [lt] "=&s"(ltm), [le] "=&s"(lem), [t0] "=&s"(t0), [cm] "=&s"(cmw)
        : [k] "v"(k), [hi] "s"(hi), [tg] "s"(target), [cmin] "s"(cm)
        : "scc");
#endif
    return kp;
}

// The same over TWO keys per lane (k0 of lanes cm0, k1 of lanes cm1), exact: the key kp with at most `target` keys below it and more than
// `target` not above it; the four masks of the last round are returned.
__device__ __forceinline__ uint32_t wave_kth_u32x2(uint32_t k0, uint32_t k1, u64 cm0, u64 cm1, int target, u64 &lt0, u64 &lt1, u64 &le0,
                                                   u64 &le1) {
    uint32_t kp;
    int t0, t1;
    u64 c0, c1;
    asm("s_mov_b64 %[c0], %[i0]\n\t"
        "s_mov_b64 %[c1], %[i1]\n\t"
        "1:\n\t"
        "s_cmp_lg_u64 %[c0], 0\n\t"
        "s_cbranch_scc0 4f\n\t"
        "s_ff1_i32_b64 %[t0], %[c0]\n\t"
        "v_readlane_b32 %[kp], %[k0], %[t0]\n\t"
        "s_branch 5f\n\t"
        "4:\n\t"
        "s_ff1_i32_b64 %[t0], %[c1]\n\t"
        "v_readlane_b32 %[kp], %[k1], %[t0]\n\t"
        "5:\n\t"
        "s_nop 1\n\t"
        "v_cmp_gt_u32_e64 %[lt0], %[kp], %[k0]\n\t"
        "v_cmp_gt_u32_e64 %[lt1], %[kp], %[k1]\n\t"
        "s_bcnt1_i32_b64 %[t0], %[lt0]\n\t"
        "s_bcnt1_i32_b64 %[t1], %[lt1]\n\t"
        "s_add_i32 %[t0], %[t0], %[t1]\n\t"
        "s_cmp_gt_u32 %[t0], %[tg]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "v_cmp_ge_u32_e64 %[le0], %[kp], %[k0]\n\t"
        "v_cmp_ge_u32_e64 %[le1], %[kp], %[k1]\n\t"
        "s_bcnt1_i32_b64 %[t0], %[le0]\n\t"
        "s_bcnt1_i32_b64 %[t1], %[le1]\n\t"
        "s_add_i32 %[t0], %[t0], %[t1]\n\t"
        "s_cmp_gt_u32 %[t0], %[tg]\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_andn2_b64 %[c0], %[c0], %[le0]\n\t"
        "s_andn2_b64 %[c1], %[c1], %[le1]\n\t"
        "s_branch 1b\n\t"
        "2:\n\t"
        "s_and_b64 %[c0], %[c0], %[lt0]\n\t"
        "s_and_b64 %[c1], %[c1], %[lt1]\n\t"
        "s_branch 1b\n\t"
        "3:"
        : [kp] "=&s"(kp), [lt0] "=&s"(lt0), [lt1] "=&s"(lt1), [le0] "=&s"(le0), [le1] "=&s"(le1), [t0] "=&s"(t0), [t1] "=&s"(t1),
          [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [k0] "v"(k0), [k1] "v"(k1), [tg] "s"(target), [i0] "s"(cm0), [i1] "s"(cm1)
        : "scc");
    return kp;
}

// d + bit `lane` of the wave-uniform mask m, as ONE instruction (add with carry-in; the compiler's own form of d + (s ? 1 : 0) is a
// select and an add)
__device__ __forceinline__ int add_lane_bit(int d, u64 m) {
    int r;
    u64 carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=&s"(carry_out) : "v"(d), "s"(m));
    return r;
}

// The cnt smallest of the 32-bit keys `ko` (one per lane of `valid`, more than cnt of them; the other lanes hold 0xffffffff):
// `has` per lane, the mask as the return value.  Equal keys across the boundary go to the lowest lanes.
__device__ __forceinline__ u64 wave_select_mask(uint32_t ko, u64 valid, int cnt, bool &has) {
    u64 ltm, lem;
    const uint32_t kp = wave_kth_u32(ko, valid, cnt - 1, 0, ltm, lem);
    if (__popcll(lem) == cnt) {
        has = ko <= kp;
        return lem;
    }
    const u64 eq = lem & ~ltm;
    const int need = cnt - __popcll(ltm);
    has = (ko < kp) | ((ko == kp) & (mbcnt64(eq) < need));      // (exact ties only)
    return __ballot(has);
}

// How the wave holds its candidates: kLaneMajor -- position p[i] = VPL * lane + i (a lane holds VPL consecutive candidates: the
// loads of stage 0 are 16-byte pieces per lane); kSlotMajor -- p[i] = 64 * i + lane (a 16-lane group shares a table row: the
// gathers of the level-0 combine); kAnyOrder -- anything (the merge of a chunked selection).  Either way the list is handed
// over in ascending position.
enum { kLaneMajor = 0, kSlotMajor = 1, kAnyOrder = 2 };
template <int VPL, int LAYOUT = kLaneMajor>
__device__ __forceinline__ int wave_select_set(const float (&v)[VPL], const int (&p)[VPL], int cnt, int M,
                                               u64 *lds /* kSelectLdsU64 per wave */, bool &has, int &dst, float &out_v, int &out_p) {
    const int lane = lane_id();
    if (cnt == 1) {  // plain arg-min
        wave_select<VPL>(v, p, 1, M, out_v, out_p);
        has = lane == 0;
        dst = 0;
        return 1;
    }
    if constexpr (VPL == 1 && LAYOUT != kAnyOrder) {
        const bool cand = p[0] != kBigPos;
        const u64 valid = __ballot(cand);
        const uint32_t ko = cand ? ord32(v[0]) : 0xffffffffu;
        u64 selm = valid;
        has = cand;
        if (__popcll(valid) > cnt) selm = wave_select_mask(ko, valid, cnt, has);
        dst = mbcnt64(selm);
        out_v = v[0];
        out_p = p[0];
        return __popcll(selm);
    }
    if constexpr (VPL > 1 && LAYOUT != kAnyOrder) {
        // (every slot holds a candidate: the callers with several keys per lane have no empty positions)
        float lm = v[0];
#pragma unroll
        for (int i = 1; i < VPL; ++i) lm = __builtin_fminf(lm, v[i]);
        u64 ltm, lem;
        const float T0v = unord32(wave_kth_u32(ord32(lm), ~0ull, cnt - 1, MCQ_SEL_WIN, ltm, lem));
        // survivors: at least cnt (one in each of >= cnt lanes), usually about 1.5 cnt
        u64 m[VPL];
        int c = 0;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            m[i] = __ballot(v[i] <= T0v);
            c += __popcll(m[i]);
        }
        if (c >= cnt && c <= 64) {
            // score and position of survivor number d (in position order) go to lo[d] and lo[128 + d]: one ds_write2_b32 from
            // the registers they are in
            uint32_t *lo = reinterpret_cast<uint32_t *>(lds);
            if constexpr (LAYOUT == kLaneMajor) {
                int d = 0;
#pragma unroll
                for (int i = 0; i < VPL; ++i) d = mbcnt64(m[i], d);        // survivors in the lanes below this one
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    if (v[i] <= T0v) { lo[d] = __float_as_uint(v[i]); lo[128 + d] = (uint32_t)p[i]; }
                    if (i + 1 < VPL) d = add_lane_bit(d, m[i]);
                }
            } else {
                int base = 0;                                              // survivors in the slots before this one (wave-uniform)
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const int d = mbcnt64(m[i], base);
                    if (v[i] <= T0v) { lo[d] = __float_as_uint(v[i]); lo[128 + d] = (uint32_t)p[i]; }
                    base += __popcll(m[i]);
                }
            }
            wave_lds_fence();
            const uint32_t ev = lo[lane], ep = lo[128 + lane];              // (lanes >= c read what an earlier selection left: unused)
            wave_lds_fence();                                              // (read before a later selection writes here)
            out_v = __uint_as_float(ev);
            out_p = (int)ep;
            const bool in = lane < c;
            u64 selm = (c == 64) ? ~0ull : ((1ull << c) - 1ull);
            has = in;
            if (c != cnt) selm = wave_select_mask(in ? ord32(out_v) : 0xffffffffu, selm, cnt, has);
            dst = mbcnt64(selm);
            return cnt;
        }
        if (c > 64 && c <= 128) {
            // The survivors cluster (a lane's keys share a table row: up to all of them lie below the bound): TWO per lane.  Survivor d,
            // in position order, goes to slot d / 64 of lane d % 64; an exact quickselect over both slots; the cnt selected are then
            // brought to one per lane, in the same order, through the same scratch.
            uint32_t *lo = reinterpret_cast<uint32_t *>(lds);
            if constexpr (LAYOUT == kLaneMajor) {
                int d = 0;
#pragma unroll
                for (int i = 0; i < VPL; ++i) d = mbcnt64(m[i], d);
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    if (v[i] <= T0v) { lo[d] = __float_as_uint(v[i]); lo[128 + d] = (uint32_t)p[i]; }
                    if (i + 1 < VPL) d = add_lane_bit(d, m[i]);
                }
            } else {
                int base = 0;
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const int d = mbcnt64(m[i], base);
                    if (v[i] <= T0v) { lo[d] = __float_as_uint(v[i]); lo[128 + d] = (uint32_t)p[i]; }
                    base += __popcll(m[i]);
                }
            }
            wave_lds_fence();
            const uint32_t v0 = lo[lane], v1 = lo[64 + lane], p0 = lo[128 + lane], p1 = lo[192 + lane];
            wave_lds_fence();
            const bool in1 = lane + 64 < c;
            const u64 cm1 = (c == 128) ? ~0ull : ((1ull << (c - 64)) - 1ull);
            const uint32_t k0 = ord32(__uint_as_float(v0)), k1 = in1 ? ord32(__uint_as_float(v1)) : 0xffffffffu;
            u64 lt0, lt1, le0, le1;
            const uint32_t kp = wave_kth_u32x2(k0, k1, ~0ull, cm1, cnt - 1, lt0, lt1, le0, le1);
            bool h0 = k0 <= kp, h1 = k1 <= kp;
            u64 s0 = le0, s1 = le1;
            if (__popcll(le0) + __popcll(le1) != cnt) {      // equal scores across the boundary: the lowest positions fill the list
                const u64 eq0 = le0 & ~lt0, eq1 = le1 & ~lt1;
                const int need = cnt - __popcll(lt0) - __popcll(lt1);
                h0 = (k0 < kp) | ((k0 == kp) & (mbcnt64(eq0) < need));
                h1 = (k1 < kp) | ((k1 == kp) & (mbcnt64(eq1, __popcll(eq0)) < need));
                s0 = __ballot(h0);
                s1 = __ballot(h1);
            }
            const int d0 = mbcnt64(s0), d1 = mbcnt64(s1, __popcll(s0));
            if (h0) { lo[d0] = v0; lo[128 + d0] = p0; }
            if (h1) { lo[d1] = v1; lo[128 + d1] = p1; }
            wave_lds_fence();
            const uint32_t ev = lo[lane], ep = lo[128 + lane];
            wave_lds_fence();
            out_v = __uint_as_float(ev);
            out_p = (int)ep;
            has = lane < cnt;
            dst = lane;
            return cnt;
        }
    }
    // general form: unique 64-bit keys (score image || position), an exact quickselect over all slots -- candidate sets as
    // wave-uniform masks, one per slot; the pivot is the first candidate of the lowest slot that has one
    u64 key[VPL];
    u64 cmask[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bool cand = p[i] != kBigPos;
        key[i] = cand ? (((u64)ord32(v[i]) << 32) | (uint32_t)p[i]) : kKeyMax;
        cmask[i] = __ballot(cand);
    }
    const int target = cnt - 1;
    u64 T = 0;
    for (;;) {                // every round removes at least the pivot from the candidates: <= 64 * VPL rounds
        u64 kp = kKeyMax;
        int ps = -1, pl = 0;
#pragma unroll
        for (int i = VPL - 1; i >= 0; --i)
            if (cmask[i] != 0) { ps = i; pl = __ffsll((long long)cmask[i]) - 1; }
        if (ps < 0) { T = kKeyMax - 1; break; }      // fewer than cnt candidates: all of them (their keys are below kKeyMax - 1)
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
    // the selected keys, one per lane (at most cnt <= 64 of them), in position order where the layout gives one
    uint32_t *lo = reinterpret_cast<uint32_t *>(lds);
    int d = 0, nsel = 0;
    if constexpr (LAYOUT == kSlotMajor) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool s = key[i] <= T;
            const u64 mi = __ballot(s);
            d = mbcnt64(mi, nsel);
            if (s && d < 64) { lo[d] = __float_as_uint(v[i]); lo[128 + d] = (uint32_t)p[i]; }
            nsel += __popcll(mi);
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const u64 mi = __ballot(key[i] <= T);
            d = mbcnt64(mi, d);
            nsel += __popcll(mi);
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bool s = key[i] <= T;
            if (s && d < 64) { lo[d] = __float_as_uint(v[i]); lo[128 + d] = (uint32_t)p[i]; }
            d += s ? 1 : 0;
        }
    }
    wave_lds_fence();
    has = lane < nsel;
    const uint32_t ev = lo[lane], ep = lo[128 + lane];
    wave_lds_fence();
    out_v = __uint_as_float(ev);
    out_p = (int)ep;
    dst = lane;
    if constexpr (LAYOUT == kAnyOrder) {
        // the list is in ascending position whatever the lanes' order: rank by position among the selected
        const int mine = has ? out_p : kBigPos;
        int r = 0;
        for (int j = 0; j < nsel; ++j) r += (__builtin_amdgcn_readlane(mine, j) < mine) ? 1 : 0;
        dst = r;
    }
    return nsel;
}

// Test hook: one wave selects the `cnt` smallest of M = 64 * VPL scores (position = index) with wave_select_set; entry j of the
// list (ascending position) goes to out[j]; a list that runs out of candidates is padded with (INF, M - 1) as the oracle pads
// it.  Lets the tests drive every path of the selection with adversarial inputs (ties, all survivors in a few lanes, more
// survivors than one per lane).
template <int VPL, int LAYOUT = kLaneMajor>
__global__ void __launch_bounds__(64)
k_test_select(const float *__restrict__ scores, int cnt, float *__restrict__ out_v, int *__restrict__ out_p) {
    __shared__ u64 scratch[kSelectLdsU64];
    const float *sc = scores + (size_t)blockIdx.x * 64 * VPL;
    float v[VPL];
    int p[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        p[i] = (LAYOUT == kSlotMajor) ? 64 * i + lane_id() : VPL * lane_id() + i;
        v[i] = sc[p[i]];
    }
    bool has;
    int dst, op;
    float ov;
    const int nsel = wave_select_set<VPL, LAYOUT>(v, p, cnt, 64 * VPL, scratch, has, dst, ov, op);
    if (has) {
        out_v[(size_t)blockIdx.x * 64 + dst] = ov;
        out_p[(size_t)blockIdx.x * 64 + dst] = op;
    }
    if (lane_id() >= nsel && lane_id() < cnt) {
        out_v[(size_t)blockIdx.x * 64 + lane_id()] = INFINITY;
        out_p[(size_t)blockIdx.x * 64 + lane_id()] = 64 * VPL - 1;
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
