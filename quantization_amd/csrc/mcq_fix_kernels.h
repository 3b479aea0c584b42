// mcq_fix_kernels.h -- the three inner-product tables of the path (logits :277-279, x.C and the Gram matrix of the
// centers: oracle/mcq_oracle.c "fixdot") as EXACT fixed-point products on the i8 matrix cores of gfx950.
//
// A row v is held as 30-bit fixed point against its own largest magnitude (row exponent e: max|v| < 2^e,
// q = rint(v * 2^(30-e))) and q as four signed 8-bit limbs, q = l0*2^24 + l1*2^16 + l2*2^8 + l3.  The product of two rows is
//     T_s = sum_k sum_{i+j=s} la_i[k] * lb_j[k]          s = 0..3: ten limb products, four i32 accumulator sets
//     t   = fma(T_0, 2^24, fma(T_1, 2^16, fma(T_2, 2^8, (float)T_3)))     (int -> float conversions round to nearest even)
//     fixdot = ldexp(t, ea + eb - 36)
// The T_s are exact integers, so nothing depends on the order the matrix cores add in; products of weight 2^-32 and
// below (i + j >= 4) are dropped.  v_mfma_i32_32x32x32_i8 runs at 32x the rate of the fp32 MFMA; ten of them replace one
// fp32 product step.
//
// Layout of a limb matrix with R rows (R a multiple of 128) and Dq columns (a multiple of 128): one "plane" per
// (16-column chunk c, limb l), each plane R x 16 bytes:  byte ((c * 4 + l) * R + row) * 16 + (k % 16).  A 128-row tile of
// one plane is 2 KB contiguous: the GEMM brings it into LDS with two 1 KB LDS-DMA loads and reads MFMA fragments
// (16 consecutive k of one row) straight out of it with ds_read_b128, 16 lanes per 256 contiguous bytes.
#pragma once
#include "mcq_kernels.h"

namespace mcq {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kFixTile = 128;          // rows per tile, both operands
constexpr int kFixColPad = 128;        // columns are padded to a multiple of this (four k steps of 32)
constexpr int kFixRing = 4;            // LDS ring slots, one k step (32 columns) each
constexpr int kFixSlot = 32768;        // 2 operands x 8 planes x 2 KB
constexpr int kFixOperand = 16384;
constexpr int kFixInfo = kFixRing * kFixSlot;      // per-tile row / column data after the ring (4 KB)
constexpr int kFixLds = kFixInfo + 4096;

__host__ __device__ inline long fix_round_rows(long r) { return (r + kFixTile - 1) / kFixTile * kFixTile; }
__host__ __device__ inline int fix_round_cols(int d) { return (d + kFixColPad - 1) / kFixColPad * kFixColPad; }
// bytes of the planes / of the exponents of an R x D matrix
__host__ __device__ inline size_t fix_plane_bytes(long R, int D) { return (size_t)fix_round_rows(R) * fix_round_cols(D) * 4; }

// ------------------------------------------------------------------ rows -> limb planes
__device__ __forceinline__ int fix_q(float v, int e) {
    float s = ldexpf(v, 30 - e);
    s = fminf(fmaxf(s, -1073741824.0f), 1073741824.0f);
    return (int)rintf(s);
}

// Workgroup = RW rows (4 is what ships: one row per wave), wave w takes RW / 4 of them.  Pass 1: the row
// exponent (and, for frames, |x|^2 as the sum of squares of the path is formed: lane l adds the float4 groups l, l + 64,
// ... then the butterfly); rows of up to 1,024 columns stay in registers for pass 2.  Pass 2, per block of 512 columns:
// limbs to LDS as [plane][row][16 bytes], then runs of RW x 16 bytes (RW rows of one plane) to the planes.
// src: fp32 rows of `ld` floats (or fp16 rows of `ld` halves when xh), D valid columns; rows >= R are written as zeros.
// bias_src / bias_dst: optional copy of R floats riding along (the classifier's bias into `prepared`).
struct FixRowsArgs {
    const float *src;
    int xh;
    long R, Rp;
    int D;
    long ld;
    int Dq;
    int8_t *planes;
    int *exps;
    float *xx;
    const float *bias_src;
    float *bias_dst;
    // centering (oracle "CENTERING": every table of the search is formed from rows / frames with the codebook means taken
    // out): row r is written as src[r] - sub[(r / sub_per) * sub_ld] (one fp32 subtraction per element; sub_per == 0: every
    // row takes sub[0 .. D)); planes / exps / xx receive the CENTERED row.
    const float *sub;
    long sub_per, sub_ld;
    // optional: dot_out[r] = fixdot(row r, dot_vec[0 .. D)) -- the exact fixed-point product of the path, formed from the limbs this
    // kernel has in its registers anyway (the classifier rows against the data mean: what centering the frame takes out of a logit)
    const float *dot_vec;
    float *dot_out;
};

template <int RW, bool DOT>      // DOT: also the rows' fixed-point products with a.dot_vec (the classifier rows of mcq_prepare)
__device__ __forceinline__ void fix_rows_body(const FixRowsArgs &a, unsigned bid) {
    const float *__restrict__ src = a.src;
    const int xh = a.xh, D = a.D, Dq = a.Dq;
    const long R = a.R, Rp = a.Rp, ld = a.ld;
    float *__restrict__ xx = a.xx;
    const float *__restrict__ bias_src = a.bias_src;
    float *__restrict__ bias_dst = a.bias_dst;
    const float *__restrict__ sub = a.sub;
    int8_t *__restrict__ planes = a.planes;
    constexpr int RPW = RW / 4;
    __shared__ __attribute__((aligned(16))) unsigned tile[128 * RW * 4];      // [plane of the block][row][4 words]
    __shared__ int es[RW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)bid * RW;
    if (bias_src && tid < RW && row0 + tid < R) bias_dst[row0 + tid] = bias_src[row0 + tid];
    const _Float16 *srch = reinterpret_cast<const _Float16 *>(src);
    const bool vec_ok = ((ld & 3) == 0) && ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & (xh ? 7 : 15)) == 0);
    auto load4 = [&](long row, int q) -> f32x4 {      // float4 group q of a row, zero past D
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row >= R || 4 * q >= D) return v;
        if (vec_ok) return xh ? load_h4(srch + row * ld + 4 * q) : *reinterpret_cast<const f32x4 *>(src + row * ld + 4 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * q + c < D) v[c] = xh ? (float)srch[row * ld + 4 * q + c] : src[row * ld + 4 * q + c];
        return v;
    };
    // group q of a D-vector (zeros past D): what is taken out of a row (`sub`; nothing for the padding rows, which stay zero rows)
    // or what the rows are multiplied with (`dot_vec`)
    auto vec4 = [&](const float *p, int q) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (4 * q >= D) return v;
        if (((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) return *reinterpret_cast<const f32x4 *>(p + 4 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * q + c < D) v[c] = p[4 * q + c];
        return v;
    };
    auto sub4 = [&](long row, int q) -> f32x4 {
        if (!sub || row >= R) return f32x4{0.f, 0.f, 0.f, 0.f};
        return vec4(sub + (a.sub_per ? (row / a.sub_per) * a.sub_ld : 0), q);
    };
    // the exponent of dot_vec (every wave forms it: max |.| over the vector)
    int ev = 0;
    if (DOT && a.dot_vec) {
        float mv = 0.f;
        for (int q = lane; q < (D + 3) / 4; q += 64) {
            const f32x4 v = vec4(a.dot_vec, q);
#pragma unroll
            for (int c = 0; c < 4; ++c) mv = fmaxf(mv, fabsf(v[c]));
        }
        for (int s = 32; s >= 1; s >>= 1) mv = fmaxf(mv, __shfl_xor(mv, s, 64));
        const int be = (int)((__float_as_uint(mv) >> 23) & 0xff);
        ev = (be < 1 ? 1 : be) - 126;
    }
    const bool cached = (RW == 4) && D <= 1024;      // (16-row workgroups measured slower with the rows held, 0.126 vs 0.083 ms, and slower than 4-row ones either way)
    f32x4 cache[RW == 4 ? RPW : 1][4];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const long row = row0 + RPW * wave + rr;
        float m = 0.f, pe = 0.f;
        if (cached) {
            // (the row and what is taken out of it are requested together, then subtracted: written as load - load per group the
            // compiler waited for each pair in turn, 61 -> 71 us per 65,536 frames)
            f32x4 raw[4], sv4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = load4(row, lane + 64 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) sv4[j] = sub4(row, lane + 64 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) cache[RW == 4 ? rr : 0][j] = sub ? raw[j] - sv4[j] : raw[j];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    m = fmaxf(m, fabsf(cache[RW == 4 ? rr : 0][j][c]));
                    pe = fmaf(cache[RW == 4 ? rr : 0][j][c], cache[RW == 4 ? rr : 0][j][c], pe);      // (groups past D are zeros: fmaf(0, 0, pe) == pe)
                }
        } else {
            for (int q = lane; q < (D + 3) / 4; q += 64) {
                const f32x4 raw = load4(row, q);
                const f32x4 v = sub ? raw - sub4(row, q) : raw;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    m = fmaxf(m, fabsf(v[c]));
                    pe = fmaf(v[c], v[c], pe);
                }
            }
        }
        for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
        if (xx) pe = wave_sum_butterfly(pe);
        if (lane == 0) {
            const int be = (int)((__float_as_uint(m) >> 23) & 0xff);
            const int e = (be < 1 ? 1 : be) - 126;
            es[RPW * wave + rr] = e;
            if (row < Rp) a.exps[row] = e;
            if (xx && row < R) xx[row] = pe;
        }
    }
    __syncthreads();
    int T[RPW][4];      // limb-product sums of the rows against dot_vec (oracle: fixdot), this lane's columns
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int i = 0; i < 4; ++i) T[rr][i] = 0;
    auto limbs4 = [&](const f32x4 &v, int e, unsigned (&w)[4]) {      // four columns -> one word per limb (most significant first)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int r = fix_q(v[c], e);
#pragma unroll
            for (int i = 3; i >= 1; --i) {
                const int l = (int)(int8_t)(r & 0xff);
                w[i] |= (unsigned)(l & 0xff) << (8 * c);
                r = (r - l) >> 8;
            }
            w[0] |= (unsigned)(r & 0xff) << (8 * c);
        }
    };
    for (int c0 = 0; c0 < Dq; c0 += 512) {
        const int ncol = (Dq - c0 < 512) ? Dq - c0 : 512;          // columns of this block (a multiple of 128)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int rl = RPW * wave + rr;
            const int e = es[rl];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = lane + 64 * j;
                if (q >= ncol / 4) continue;
                f32x4 v;
                if (cached) v = cache[RW == 4 ? rr : 0][((c0 >> 9) * 2 + j) & 3];
                else v = sub ? load4(row0 + rl, c0 / 4 + q) - sub4(row0 + rl, c0 / 4 + q) : load4(row0 + rl, c0 / 4 + q);
                unsigned w[4];
                limbs4(v, e, w);
                if (DOT && a.dot_vec) {
                    unsigned mw[4];
                    limbs4(vec4(a.dot_vec, c0 / 4 + q), ev, mw);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; i + jj < 4; ++jj) T[rr][i + jj] = __builtin_amdgcn_sdot4((int)w[i], (int)mw[jj], T[rr][i + jj], false);
                }
                const int chunk = q >> 2, word = q & 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[((chunk * 4 + i) * RW + rl) * 4 + word] = w[i];
            }
        }
        __syncthreads();
        const int nplanes = ncol / 16 * 4;
        for (int p = tid / RW; p < nplanes; p += 256 / RW) {
            const int rl = tid % RW;
            if (row0 + rl < Rp) {
                const i32x4 v = *reinterpret_cast<const i32x4 *>(&tile[(p * RW + rl) * 4]);
                *reinterpret_cast<i32x4 *>(planes + (((long)(c0 / 16) * 4 + p) * Rp + row0 + rl) * 16) = v;
            }
        }
        __syncthreads();
    }
    if (DOT && a.dot_vec) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            int t4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int t = T[rr][i];
                for (int s = 32; s >= 1; s >>= 1) t += __shfl_xor(t, s, 64);      // exact integers: no order to speak of
                t4[i] = t;
            }
            const long row = row0 + RPW * wave + rr;
            if (lane == 0 && row < R) {
                float tt = (float)t4[3];
                tt = __builtin_fmaf((float)t4[2], 256.0f, tt);
                tt = __builtin_fmaf((float)t4[1], 65536.0f, tt);
                tt = __builtin_fmaf((float)t4[0], 16777216.0f, tt);
                a.dot_out[row] = ldexpf(tt, es[RPW * wave + rr] + ev - 36);
            }
        }
    }
}

template <int RW>
__global__ void __launch_bounds__(256) k_fix_rows(const FixRowsArgs a) { fix_rows_body<RW, false>(a, blockIdx.x); }

// two matrices in one launch (the centers and the classifier rows of mcq_prepare: one launch boundary less per trainer step)
template <int RW>
__global__ void __launch_bounds__(256) k_fix_rows2(const FixRowsArgs a, const FixRowsArgs b, unsigned blocks_a) {
    if (blockIdx.x < blocks_a) fix_rows_body<RW, false>(a, blockIdx.x);
    else fix_rows_body<RW, true>(b, blockIdx.x - blocks_a);
}

// ------------------------------------------------------------------ the GEMM
// out-of-kernel description of what the epilogue does with a tile of t values
enum { FG_STORE = 0, FG_LOGITS = 1 };

struct FixGemm {
    const int8_t *A, *B;           // limb planes: A rows are the tile rows (M), B rows the tile columns (N)
    const int *ea, *eb;            // row exponents
    long RA, RB;                   // padded row counts (multiples of 128)
    long M, N;                     // valid rows / columns
    int Dq;                        // padded inner dimension
    int walk_rows;                 // 0: an XCD owns row tiles and walks the column tiles (A streams, B is the table);
                                   // 1: the other way round (B streams)
    // FG_STORE: out[row * ldo + col] = fixdot
    float *out;
    long ldo;
    // FG_LOGITS: rows are (codebook, entry) pairs, columns are frames; value = fixdot * lscale + bias[row];
    //            (value = (fixdot + wmu[row]) * lscale + bias[row]: see wmu)
    //            logits[col * ldo + row] = value when logits != nullptr; idx[col * ncb + codebook] = first arg max
    const float *bias;
    const float *wmu;              // fixdot(data mean, W[row]): the frames of this product are centered (x - mean), the logit is (t + wmu) * lscale + bias
    const float *lscale_ptr;       // device scalar, or nullptr: lscale
    float lscale;
    float *logits;
    void *idx;                     // uint8 [N cols][ncb], or uint16 when idx_wide (codebooks of more than 256 entries)
    int idx_wide;
    int K, ncb;
};

__device__ __forceinline__ void fg_store_idx(const FixGemm &g, long pos, long entry) {
    if (g.idx_wide) static_cast<uint16_t *>(g.idx)[pos] = (uint16_t)entry;
    else static_cast<uint8_t *>(g.idx)[pos] = (uint8_t)entry;
}

// Persistent workgroups (one per CU: 132 KB of LDS): workgroup w takes the tiles w, w + grid, ... of an XCD-aware order and
// runs their k steps as ONE stream through the LDS ring -- the first steps of the next tile are in flight while this one
// finishes and stores.  EIGHT waves, two per SIMD, each a 64 x 32 corner of the 128 x 128 tile (2 MFMA tiles x 4 accumulator
// sets = 128 of its 256 registers).  A k step of a wave: 20 MFMAs against 12 ds_read_b128 (the fragments of the next step)
// and 4 LDS-DMA pieces of 1 KB (the step four ahead), dealt in four groups.  With one wave per SIMD (four waves of 64 x 64,
// the first version) a wave's DMA issue -- tens of cycles per piece in its in-order stream -- its fragment reads and its
// waits left the matrix pipe idle: 0.72-0.75 ms per product at 65,536 x 2,048 x 512; two waves cover each other: 0.58.
//
// vmcnt bookkeeping: a wave's pieces of stage s are requested in step s - 4; before the fragment reads of stage s + 1 (the
// sync of step s) it waits until only its pieces of the two later stages are outstanding (loads complete in the order they
// were issued; the stores and the few extra pieces of a tile boundary only make that count conservative), then the
// barrier makes every wave's pieces visible.  The row exponents and biases the epilogue needs reach LDS the same way (six
// 256-byte DMA pieces per tile, requested when the tile starts).  Measured and dropped: draining the queue before the
// epilogue's stores so that the next tile's first syncs need no count (0.566 against 0.549 ms).
template <int MODE>
__global__ void __launch_bounds__(512)
k_fgemm(const FixGemm g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;          // 64-row half, 32-column quarter of the tile
    const int r32 = lane & 31, kh = lane >> 5;
    const long MT = g.RA / kFixTile, NT = g.RB / kFixTile;
    // K = 256 logits: the two row tiles of a codebook are consecutive tiles of one workgroup (running arg max in registers)
    const int H = (MODE == FG_LOGITS && g.K > kFixTile) ? g.K / kFixTile : 1;
    const long big = g.walk_rows ? NT : MT, small_units = (g.walk_rows ? MT : NT) / H;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    auto tile_of = [&](long t, long &m0, long &n0) -> bool {      // t-th tile of this workgroup
        const long unit = t / H, half = t % H;
        const long idx = slot + unit * per_xcd;
        const long bt = (idx / small_units) * 8 + xcd, st = (idx % small_units) * H + half;
        m0 = (g.walk_rows ? st : bt) * kFixTile;
        n0 = (g.walk_rows ? bt : st) * kFixTile;
        return bt < big;
    };
    i32x16 acc[2][4];
    const int nst = g.Dq / 32;
    // this wave's 4 pieces of a stage (1 KB each): waves 0..3 bring operand A, 4..7 operand B; bit 1 of the wave picks the
    // 16-column chunk, bit 0 the 64-row half, g4 = 0..3 the limb plane
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int op = wu >> 2, chunk = (wu >> 1) & 1, half = wu & 1;
    const long R = op ? g.RB : g.RA;
    const long pl = R * 16, sstride = 8 * R * 16;
    const unsigned dbase = (unsigned)(size_t)smem + op * kFixOperand + (4 * chunk) * 2048 + half * 1024;
    const unsigned voff = lane * 16, voff4 = lane * 4;
    auto base_of = [&](long m0, long n0) { return (op ? g.B + n0 * 16 : g.A + m0 * 16) + (long)(4 * chunk) * R * 16 + half * 1024; };
    // (m0 is written without being declared clobbered -- the compiler rejects it as a reserved register; nothing else in
    // this kernel uses m0: LDS instructions need no m0 on gfx9+, and the kernel has no LDS-DMA builtin, movrel or GWS)
    auto issue1 = [&](const int8_t *p, int st, int g4) {
        const int8_t *pg = p + g4 * pl;
        const unsigned d = dbase + (st % kFixRing) * kFixSlot + g4 * 2048;
        asm volatile(
            "s_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[vo], %[p]\n\t"
            :
            : [vo] "v"(voff), [p] "s"(pg), [d] "s"(d)
            : "memory");
    };
    // the tile's row exponents ea (waves 0, 1), column exponents eb (2, 3) and, for the logits, row biases (4, 5) and the rows'
    // products with the data mean (6, 7) -> info: 64 words per piece; [0..127] ea, [128..255] eb, [256..383] bias, [384..639]
    // arg-max exchange, [640..767] wmu.  (Rows past M read whatever follows the bias / wmu inside `prepared`: they never reach an
    // output.)
    auto issue_info = [&](long m0, long n0) {
        if (wu < (MODE == FG_LOGITS ? 8 : 4)) {
            const void *src = wu < 2 ? static_cast<const void *>(g.ea + m0 + 64 * wu)
                                     : (wu < 4 ? static_cast<const void *>(g.eb + n0 + 64 * (wu - 2))
                                               : (wu < 6 ? static_cast<const void *>(g.bias + m0 + 64 * (wu - 4))
                                                         : static_cast<const void *>(g.wmu + m0 + 64 * (wu - 6))));
            const unsigned d = (unsigned)(size_t)smem + kFixInfo + 256 * (wu < 6 ? wu : wu + 4);
            asm volatile(
                "s_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                "global_load_lds_dword %[vo], %[p]\n\t"
                :
                : [vo] "v"(voff4), [p] "s"(src), [d] "s"(d)
                : "memory");
        }
    };
    long m0, n0, m0n = 0, n0n = 0;
    if (!tile_of(0, m0, n0)) return;
    const int8_t *pcur = base_of(m0, n0), *pnext = pcur;
    auto step = [&](int st, const i32x4 (&a)[2][4], const i32x4 (&b)[4], i32x4 (&an)[2][4], i32x4 (&bn)[4]) {
        const char *base = smem + ((st + 1) % kFixRing) * kFixSlot;
        // (after the last tile the stage four ahead does not exist: the pieces are requested all the same, from the start of
        // the current tile (pnext == pcur then) into a slot nobody reads any more -- no branch round the DMA)
        const int8_t *p = (st + 4 < nst) ? pcur + (st + 4) * sstride : pnext + (st + 4 - nst) * sstride;
        constexpr int PI[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};      // limb pairs (i, j), i + j <= 3, dealt 3, 3, 2, 2
        constexpr int PJ[10] = {0, 1, 0, 2, 1, 0, 3, 2, 1, 0};
        constexpr int LO[5] = {0, 3, 6, 8, 10};
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            issue1(p, st, g4);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                an[t][g4] = *reinterpret_cast<const i32x4 *>(base + (kh * 4 + g4) * 2048 + (64 * wm + 32 * t + r32) * 16);
            bn[g4] = *reinterpret_cast<const i32x4 *>(base + kFixOperand + (kh * 4 + g4) * 2048 + (32 * wn + r32) * 16);
#pragma unroll
            for (int q = LO[g4]; q < LO[g4 + 1]; ++q)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
                    acc[ta][PI[q] + PJ[q]] =
                        __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta][PI[q]], b[PJ[q]], acc[ta][PI[q] + PJ[q]], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // (everything requested up front instead: 0.59 against 0.53 ms)
        }
    };
    i32x4 a0[2][4], b0[4], a1[2][4], b1[4];
    issue_info(m0, n0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) issue1(pcur + q * sstride, q, g4);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int l = 0; l < 4; ++l) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            a0[t][l] = *reinterpret_cast<const i32x4 *>(smem + (kh * 4 + l) * 2048 + (64 * wm + 32 * t + r32) * 16);
        b0[l] = *reinterpret_cast<const i32x4 *>(smem + kFixOperand + (kh * 4 + l) * 2048 + (32 * wn + r32) * 16);
    }
    int *info = reinterpret_cast<int *>(smem + kFixInfo);
    float *infof = reinterpret_cast<float *>(smem + kFixInfo);
    // running arg max of a codebook that spans several row tiles: per lane, its column
    float runv = 0.f;
    int runk = 0;
    for (long t = 0;; ++t) {
        const bool has_next = tile_of(t + 1, m0n, n0n);
        pnext = has_next ? base_of(m0n, n0n) : pcur;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[a][s][v] = 0;
        // before the reads of stage st + 1: it has landed everywhere and every wave has left slot st % ring.  Two later
        // stages stay in flight
        for (int st = 0; st < nst; st += 2) {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            step(st, a0, b0, a1, b1);
            asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            step(st + 1, a1, b1, a0, b0);      // (reads past the end of the last tile hit a slot nobody uses)
        }
        // ---- epilogue.  The info words of this tile landed during its first steps (requested at its start, made visible by
        // the step barriers)
        auto value = [&](int ta, int v, int e) -> float {
            float tt = (float)acc[ta][3][v];
            tt = __builtin_fmaf((float)acc[ta][2][v], 256.0f, tt);
            tt = __builtin_fmaf((float)acc[ta][1][v], 65536.0f, tt);
            tt = __builtin_fmaf((float)acc[ta][0][v], 16777216.0f, tt);
            return ldexpf(tt, e);
        };
        // lane-dependent offsets are formed anew for every tile (hoisted out of the tile loop they would cost registers)
        int rbase = 64 * wm + 4 * kh, cbase = 32 * wn + r32;
        asm volatile("" : "+v"(rbase), "+v"(cbase));
        // the exponents (and biases) of this lane's 32 rows: eight 16-byte LDS reads up front
        i32x4 er[2][4];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int q = 0; q < 4; ++q) er[ta][q] = *reinterpret_cast<const i32x4 *>(&info[rbase + 32 * ta + 8 * q]);
        const bool full = (m0 + kFixTile <= g.M) && (n0 + kFixTile <= g.N);      // whole tile inside: no per-element guards
        const int ecol = info[128 + cbase] - 36;
        const int rows_left = (int)((g.M - m0 - rbase) > 128 ? 128 : (g.M - m0 - rbase));
        if (MODE == FG_STORE) {
            const bool col_ok = n0 + cbase < g.N;
            float *orow = g.out + (m0 + rbase) * g.ldo + n0 + cbase;
            if (full) {
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int ro = 32 * ta + 8 * (v >> 2) + (v & 3);          // row of the lane's block
                        orow[(long)ro * g.ldo] = value(ta, v, er[ta][v >> 2][v & 3] + ecol);
                    }
            } else {
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int ro = 32 * ta + 8 * (v >> 2) + (v & 3);
                        const float val = value(ta, v, er[ta][v >> 2][v & 3] + ecol);
                        if (col_ok && ro < rows_left) orow[(long)ro * g.ldo] = val;
                    }
            }
        } else {
            const float ls = g.lscale_ptr ? *g.lscale_ptr : g.lscale;
            // per lane: one column (frame), rows 64 wm + 32 ta + 8 (v >> 2) + 4 kh + (v & 3), ascending in (ta, v)
            float bv[4];          // best of the 16-row group (ta, v >> 3), this lane's 8 rows of it
            int bk[4];
            const long col = n0 + cbase;
            float *lrow = g.logits ? g.logits + col * g.ldo + m0 + rbase : nullptr;
            // the two 32-row halves one after the other, each with its own biases / mean products read where they are used: with
            // all 32 rows' worth held from the top of the epilogue (32 + 32 registers beside the 128 accumulators, the next
            // tile's first fragments and the exponents) the kernel ran out of its 256 registers (24 spilled, 0.56 -> 0.74 ms)
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const f32x4 br4 = *reinterpret_cast<const f32x4 *>(&infof[256 + rbase + 32 * ta + 8 * v4]);
                    const f32x4 wm4 = *reinterpret_cast<const f32x4 *>(&infof[640 + rbase + 32 * ta + 8 * v4]);
                    f32x4 q4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int v = 4 * v4 + c;
                        const int rl = rbase + 32 * ta + 8 * v4 + c;
                        const float val = __fadd_rn(__fmul_rn(__fadd_rn(value(ta, v, er[ta][v4][c] + ecol), wm4[c]), ls), br4[c]);
                        q4[c] = val;
                        const int gi = 2 * ta + (v4 >> 1);
                        if ((v4 & 1) == 0 && c == 0) { bv[gi] = val; bk[gi] = rl; }
                        else if (val > bv[gi]) { bv[gi] = val; bk[gi] = rl; }
                    }
                    if (lrow && (full || (col < g.N && 32 * ta + 8 * v4 < rows_left)))
                        *reinterpret_cast<f32x4 *>(lrow + 32 * ta + 8 * v4) = q4;
                }
            }
            if (g.idx) {
                // a value beats another when it is larger, or equal with the lower row
                auto better = [](float v1, int k1, float v2, int k2) { return (v1 > v2) | ((v1 == v2) & (k1 < k2)); };
                const int K = g.K;
                // the other half-wave holds the other 8 rows of every 16-row group
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const float ov = __shfl_xor(bv[gi], 32, 64);
                    const int ok = __shfl_xor(bk[gi], 32, 64);
                    if (better(ov, ok, bv[gi], bk[gi])) { bv[gi] = ov; bk[gi] = ok; }
                }
                // groups of 16 rows -> codebooks of K rows inside this wave's 64 rows (the later group wins only with a
                // strictly larger value)
                if (K >= 32) {
                    if (bv[1] > bv[0]) { bv[0] = bv[1]; bk[0] = bk[1]; }
                    if (bv[3] > bv[2]) { bv[2] = bv[3]; bk[2] = bk[3]; }
                }
                if (K >= 64 && bv[2] > bv[0]) { bv[0] = bv[2]; bk[0] = bk[2]; }
                if (K <= 64) {
                    if (kh == 0) {
                        const int span = K / 16;
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi) {
                            const long row = m0 + bk[gi];
                            if ((gi % span) == 0 && col < g.N && row < g.M) fg_store_idx(g, col * g.ncb + row / K, row % K);
                        }
                    }
                } else {
                    // K = 128: the two waves of a column meet in LDS; K = 256: and the two row tiles in registers
                    float *exv = infof + 384;
                    int *exk = info + 384 + 128;
                    if (wm == 1 && kh == 0) { exv[cbase] = bv[0]; exk[cbase] = bk[0]; }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (wm == 0 && kh == 0) {
                        const int hf = (int)(t % H);
                        float v = bv[0];
                        long k = m0 + bk[0];
                        if (better(exv[cbase], exk[cbase], bv[0], bk[0])) { v = exv[cbase]; k = m0 + exk[cbase]; }
                        if (hf > 0 && !(v > runv)) { v = runv; k = runk; }      // earlier rows win ties
                        runv = v;
                        runk = (int)k;
                        if (hf == H - 1 && col < g.N && k < g.M) fg_store_idx(g, col * g.ncb + k / K, k % K);
                    }
                }
            }
        }
        if (!has_next) break;
        // the info words are rewritten for the next tile: every wave must have read them (and every wave has waited for its
        // pieces: the next tile's first stages have landed everywhere)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        m0 = m0n;
        n0 = n0n;
        pcur = pnext;
        issue_info(m0, n0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the pieces still on their way to this workgroup's LDS
}

}  // namespace mcq
