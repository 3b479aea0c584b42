// mcq_train_kernels.h -- gfx950 kernels of QuantizerTrainer.step's parameter update
// (/root/reference/quantization/quantization.py:708-715, :722-730), the rest of SURVEY.md 8(f)-1:
//   k_wgrad_tn   dW = s * G^T x  and  db = column sums of G   (autograd of _logits, :277-279)
//   k_adam       torch.optim.Adam's update on one flat parameter / gradient / moment bucket (:712, :722-727)
//   k_loss_head  the four batch sums mcq_loss_tail consumes, from the forward kernels' partials
//   k_scales     exp(speed * scale) of the two scale parameters, on the device
//   k_centers_mean  get_data_mean() (:67-75) of the scaled centers
// Reductions have a fixed order (no float atomics): a training run is bit-reproducible.
#pragma once
#include "mcq_kernels.h"

namespace mcq {

// ------------------------------------------------------------------ dW, db
// gW[m][n] = s * sum_b G[b][m] * x[b][n]   (m: logits row n*K + k, n: feature), gb[m] = sum_b G[b][m].
// Both operands are contiguous along the OUTPUT axes and strided along the contraction axis b, so tiles go to LDS as
// they lie in memory ([b][64 + pad] rows, ds_write_b128) and the MFMA fragments are ds_read_b32 along b: lane (r, g)
// feeds row 4j + g of a 16-row block to MFMA j (rows stride 80 floats = 16 banks apart: the two half-waves of a read
// hit 32 distinct banks).  Two LDS buffers, one barrier per stage.  Workgroups of one G column slab share an XCD
// (id = nt * MT + mt).
// Workgroup = 128 (m) x 64 (n) outputs, four waves of 64 x 32 (4 x 2 MFMA tiles), ST rows of b per stage.
constexpr int kWgM = 128, kWgN = 64;
constexpr int kWgStrideA = kWgM + 16, kWgStrideB = kWgN + 16;     // floats per LDS row (stride = 16 mod 32 banks)

// The batch axis is cut into `splits` ranges (split-K): workgroup (tile, split) writes its partial tile to
// part[split][M][Nf] (and partial column sums to partb[split][M]); k_wgrad_reduce adds the splits in ascending order and
// applies s.  With tiles alone a 2048 x 512 gradient is 128 workgroups of hundreds of serial stages each; sixteen
// splits put 8 workgroups on every CU and hide the stage latency.
template <int ST>
__global__ void __launch_bounds__(256)
k_wgrad_tn(const float *__restrict__ G /*[B][M]*/, const float *__restrict__ X /*[B][Nf]*/, long B, int M, int Nf,
           long rows_per_split, float *__restrict__ gW /*part [splits][M][Nf]*/, float *__restrict__ gb /*partb [splits][M]*/) {
    __shared__ __attribute__((aligned(16))) float ldsA[2][ST * kWgStrideA];   // [buffer][row b][col m]
    __shared__ __attribute__((aligned(16))) float ldsB[2][ST * kWgStrideB];   // [buffer][row b][col n]
    __shared__ float colsum[8][kWgM];
    const int MT = (M + kWgM - 1) / kWgM, NT = (Nf + kWgN - 1) / kWgN;
    const int tile = blockIdx.x % (MT * NT), split = blockIdx.x / (MT * NT);
    const int mt = tile % MT, nt = tile / MT;
    G += split * rows_per_split * M;
    X += split * rows_per_split * Nf;
    B = (B - split * rows_per_split < rows_per_split) ? B - split * rows_per_split : rows_per_split;
    gW += (size_t)split * M * Nf;
    gb += (size_t)split * M;
    const int m0 = kWgM * mt, n0 = kWgN * nt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int r = lane & 15, g = lane >> 4;
    // staging: A tile rows of 32 float4 -> thread t moves float4 (t % 32) of rows t / 32 + 8 j; B tile rows of 16 float4 ->
    // float4 (t % 16) of rows t / 16 + 16 j
    constexpr int NA = ST / 8, NB = ST / 16;
    const int arow = tid >> 5, acol = 4 * (tid & 31);
    const int brow = tid >> 4, bcol = 4 * (tid & 15);
    const bool acol_ok = m0 + acol < M;           // M = N*K is a multiple of 16
    const bool xvec = (Nf & 3) == 0;              // rows of x are 16-byte aligned
    f32x4 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const long nst = (B + ST - 1) / ST;
    f32x4 ra[NA], rb[NB];
    auto load = [&](long st) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const long b = st * ST + arow + 8 * j;
            const long bc = b < B ? b : B - 1;
            ra[j] = (acol_ok && b < B) ? *reinterpret_cast<const f32x4 *>(G + bc * M + m0 + acol) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const long b = st * ST + brow + 16 * j;
            const long bc = b < B ? b : B - 1;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                if (xvec) {
                    if (n0 + bcol < Nf) v = *reinterpret_cast<const f32x4 *>(X + bc * Nf + n0 + bcol);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = (n0 + bcol + c < Nf) ? X[bc * Nf + n0 + bcol + c] : 0.f;
                }
            }
            rb[j] = v;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            *reinterpret_cast<f32x4 *>(&ldsA[buf][(arow + 8 * j) * kWgStrideA + acol]) = ra[j];
            csum = csum + ra[j];                    // rows b = arow (mod 8) of this thread's four columns, b ascending
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4 *>(&ldsB[buf][(brow + 16 * j) * kWgStrideB + bcol]) = rb[j];
    };
    load(0);
    store(0);
    __syncthreads();
    if (nst > 1) load(1);
    for (long st = 0; st < nst; ++st) {
        const int buf = (int)(st & 1);
        const float *A = ldsA[buf], *Bt = ldsB[buf];
#pragma unroll
        for (int j = 0; j < ST / 4; ++j) {
            float a[4], b[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = A[(4 * j + g) * kWgStrideA + 64 * wm + 16 * t + r];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = Bt[(4 * j + g) * kWgStrideB + 32 * wn + 16 * u + r];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (st + 1 < nst) store(buf ^ 1);
        __syncthreads();
        if (st + 2 < nst) load(st + 2);
    }
    // lane holds rows m = 64 wm + 16 t + 4 g + v, column n = 32 wn + 16 u + r
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = n0 + 32 * wn + 16 * u + r;
            if (n < Nf) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = m0 + 64 * wm + 16 * t + 4 * g + v;
                    if (m < M) gW[(long)m * Nf + n] = acc[t][u][v];
                }
            }
        }
    if (nt == 0) {     // column sums of this G slab: the 8 row classes are added in order
#pragma unroll
        for (int c = 0; c < 4; ++c) colsum[arow][acol + c] = csum[c];
        __syncthreads();
        if (tid < kWgM) {
            float t = colsum[0][tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) t = t + colsum[q][tid];
            if (m0 + tid < M) gb[m0 + tid] = t;
        }
    }
}

// ------------------------------------------------- dW, db on the bf16 matrix cores
// The same partial tiles as k_wgrad_tn for tile-aligned shapes (M and Nf multiples of 128: the trainer's second phase, where
// this product is the largest kernel of a step).  Every fp32 operand is cut into THREE bf16 pieces by truncation (a = a0 + a1 +
// a2 + e, |e| < 2^-24 |a|: the pieces are the top 16 bits of a, of a - a0 and of a - a0 - a1, each difference exact in fp32), and
// the six piece products of weight >= 2^-16 run as v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-grade results (the
// dropped products are below 2^-24 of a term) at a sixteenth of the fp32 MFMA's time per product.  The gradient is not part of
// the parity contract of the index search (tests bound it against the fp64 product).
// Layout: the MFMA wants eight consecutive b (the contraction axis) per lane, memory has b as the SLOW axis of both operands.  A
// thread therefore loads one column: eight consecutive b of one m (or n) with eight dword loads, each coalesced across the wave
// (lanes = consecutive columns), splits them and writes the three pieces as one ds_write_b128 per plane into [column][b] images
// (rows of 32 + 8 bf16: the ds_read_b128 of sixteen rows then fall on 64 distinct banks).  Two LDS buffers, one barrier per
// stage of 32 b: a wave splits and writes stage st + 1 beside its MFMAs of stage st, the loads run two stages ahead; 123 KB of
// LDS: one workgroup of eight waves per CU.
constexpr int kWbRow = 40;                                  // bf16 per LDS row
constexpr int kWbStage = 32;                                // b per stage
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// eight fp32 -> three planes of eight bf16 (element i in half i % 2 of dword i / 2)
__device__ __forceinline__ void bf3_split8(const float (&v)[8], u32x4 &p0, u32x4 &p1, u32x4 &p2) {
    uint32_t b0[8], b1[8], b2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        b0[i] = __float_as_uint(v[i]);
        const float r1 = v[i] - __uint_as_float(b0[i] & 0xffff0000u);
        b1[i] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(b1[i] & 0xffff0000u);
        b2[i] = __float_as_uint(r2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // the top halves of elements 2j (low) and 2j + 1 (high)
        p0[j] = __builtin_amdgcn_perm(b0[2 * j + 1], b0[2 * j], 0x07060302u);
        p1[j] = __builtin_amdgcn_perm(b1[2 * j + 1], b1[2 * j], 0x07060302u);
        p2[j] = __builtin_amdgcn_perm(b2[2 * j + 1], b2[2 * j], 0x07060302u);
    }
}

constexpr int kWbM = 128, kWbN = 128;                       // outputs per workgroup: eight waves of 64 (m) x 32 (n)

__global__ void __launch_bounds__(512)
k_wgrad_bf3(const float *__restrict__ G /*[B][M]*/, const float *__restrict__ X /*[B][Nf]*/, long B, int M, int Nf,
            long rows_per_split, float *__restrict__ gW /*part [splits][M][Nf]*/, float *__restrict__ gb /*partb [splits][M]*/) {
    __shared__ __attribute__((aligned(16))) __bf16 ldsA[2][3][kWbM * kWbRow];      // [buffer][plane][m][b]
    __shared__ __attribute__((aligned(16))) __bf16 ldsB[2][3][kWbN * kWbRow];      // [buffer][plane][n][b]
    __shared__ float colsum[4][kWbM];
    const int MT = M / kWbM, NT = Nf / kWbN;
    const int tile = blockIdx.x % (MT * NT), split = blockIdx.x / (MT * NT);
    const int mt = tile % MT, nt = tile / MT;
    G += split * rows_per_split * M;
    X += split * rows_per_split * Nf;
    B = (B - split * rows_per_split < rows_per_split) ? B - split * rows_per_split : rows_per_split;
    gW += (size_t)split * M * Nf;
    gb += (size_t)split * M;
    const int m0 = kWbM * mt, n0 = kWbN * nt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    // staging: thread t holds the octet t / 128 of a stage for column m0 + t % 128 of G and for column n0 + t % 128 of x
    const int sc = tid & 127, so = tid >> 7;
    const float *ga = G + m0 + sc, *xb = X + n0 + sc;
    float fa[8], fb[8], ga2[8], gb2[8];      // two register sets: the loads run TWO stages ahead of their use
    float csum = 0.f;
    // load(st) only ISSUES the loads of stage st (rows past the end of the split are read from its last row and dropped in
    // store(): a select on the loaded value here made the compiler wait for every load before the MFMAs of the stage in front)
    auto load = [&](long st, float (&fa)[8], float (&fb)[8]) {
        const float *gs = ga + st * kWbStage * M, *xs = xb + st * kWbStage * Nf;
        const int lim = (int)(B - st * kWbStage) - 1;              // last valid row of the stage (>= 0)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * so + i;
            const uint32_t rc = (uint32_t)(r < lim ? r : lim);
            fa[i] = gs[rc * (uint32_t)M];
            fb[i] = xs[rc * (uint32_t)Nf];
        }
    };
    auto store = [&](long st, float (&fa)[8], float (&fb)[8]) {
        const int buf = (int)(st & 1);
        const int lim = (int)(B - st * kWbStage) - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = 8 * so + i <= lim;
            fa[i] = ok ? fa[i] : 0.f;
            fb[i] = ok ? fb[i] : 0.f;
        }
        u32x4 p0, p1, p2;
        const int at = sc * kWbRow + 8 * so;
        bf3_split8(fa, p0, p1, p2);
        *reinterpret_cast<u32x4 *>(&ldsA[buf][0][at]) = p0;
        *reinterpret_cast<u32x4 *>(&ldsA[buf][1][at]) = p1;
        *reinterpret_cast<u32x4 *>(&ldsA[buf][2][at]) = p2;
        bf3_split8(fb, p0, p1, p2);
        *reinterpret_cast<u32x4 *>(&ldsB[buf][0][at]) = p0;
        *reinterpret_cast<u32x4 *>(&ldsB[buf][1][at]) = p1;
        *reinterpret_cast<u32x4 *>(&ldsB[buf][2][at]) = p2;
#pragma unroll
        for (int i = 0; i < 8; ++i) csum = csum + fa[i];      // b ascending within the thread's octet
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long nst = (B + kWbStage - 1) / kWbStage;
    // fragment addresses: lane l feeds row (column of the image) l % 32 with the octet l / 32 of a 16-b step
    const int fr = lane & 31, fo = lane >> 5;
    // the fragments of the second 16-b step are requested before the MFMAs of the first (one exposed LDS round trip per stage
    // instead of two: with the reads in front of their own MFMAs they were the largest item of the kernel, 32 of 66 us)
    auto frags = [&](int buf, int kk, bf16x8 (&a)[2][3], bf16x8 (&b)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                a[t][p] = *reinterpret_cast<const bf16x8 *>(&ldsA[buf][p][(64 * wm + 32 * t + fr) * kWbRow + 16 * kk + 8 * fo]);
            b[p] = *reinterpret_cast<const bf16x8 *>(&ldsB[buf][p][(32 * wn + fr) * kWbRow + 16 * kk + 8 * fo]);
        }
    };
    auto six = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[3]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {      // the small products first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][2], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[0], acc[t], 0, 0, 0);
        }
    };
    auto mfmas = [&](int buf) {
        bf16x8 a0[2][3], b0[3], a1[2][3], b1[3];
        frags(buf, 0, a0, b0);
        frags(buf, 1, a1, b1);
        six(a0, b0);
        six(a1, b1);
    };
    // Stage st + 1 is split and written into the OTHER buffer beside the MFMAs of stage st (its loads were issued two stages
    // back); one barrier per stage.  Stages in pairs: even ones through (fa, fb), odd ones through (ga2, gb2).  The body is
    // branch-free (a stage past the end re-reads the last one and stores zeros) so that the scheduler may deal the VALU work
    // of the split between the MFMAs.
    const long last = nst - 1;
    if (nst > 0) {
        load(0, fa, fb);
        load(nst > 1 ? 1 : last, ga2, gb2);
        store(0, fa, fb);
        load(nst > 2 ? 2 : last, fa, fb);
    }
    __syncthreads();
    for (long st = 0; st < nst; st += 2) {
        mfmas(0);
        store(st + 1, ga2, gb2);
        load(st + 3 < nst ? st + 3 : last, ga2, gb2);
        __syncthreads();
        mfmas(1);
        store(st + 2, fa, fb);
        load(st + 4 < nst ? st + 4 : last, fa, fb);
        __syncthreads();
    }
    // lane l, element r: row m = 64 wm + 32 t + 8 (r / 4) + 4 (l / 32) + r % 4, column n = 32 wn + l % 32
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 64 * wm + 32 * t + 8 * (r >> 2) + 4 * fo + (r & 3);
            gW[(long)m * Nf + n0 + 32 * wn + fr] = acc[t][r];
        }
    if (nt == 0) {     // column sums of this G slab: the four octet classes in order
        colsum[so][sc] = csum;
        __syncthreads();
        if (tid < kWbM) gb[m0 + tid] = ((colsum[0][tid] + colsum[1][tid]) + colsum[2][tid]) + colsum[3][tid];
    }
}

// gW[i] = s * (part[0][i] + part[1][i] + ...), gb likewise without the factor: splits ascending
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float *__restrict__ part, const float *__restrict__ partb, int splits, long MN, int M,
               const float *__restrict__ scale, float *__restrict__ gW, float *__restrict__ gb) {
    const float s = *scale;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    // the partials of eight splits are requested together, then added in ascending order (a plain loop became load, wait,
    // add per split: one memory round trip each, 16.6 us for sixteen splits)
    auto sum_splits = [&](const float *base, size_t stride) {
        f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < splits; q0 += 8) {
            f32x4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u < splits ? q0 + u : splits - 1;
                r[u] = *reinterpret_cast<const f32x4 *>(base + (size_t)q * stride);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + u < splits) t = (q0 + u == 0) ? r[u] : t + r[u];
        }
        return t;
    };
    if (i < MN) *reinterpret_cast<f32x4 *>(gW + i) = sum_splits(part + i, (size_t)MN) * s;       // MN is a multiple of 4 (M = N*K is a multiple of 16)
    if (i < M) *reinterpret_cast<f32x4 *>(gb + i) = sum_splits(partb + i, (size_t)M);
}

// --------------------------------------------------------------------- Adam
// torch.optim.Adam (L2 weight decay, no amsgrad), the arithmetic of its fused implementation in fp32:
//   g += wd * p;  m += (1 - beta1) * (g - m);  v = beta2 * v + (1 - beta2) * g * g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// over one flat bucket (the parameters' .data and .grad are views of p and g).  lr / bc1, 1 - beta1, 1 - beta2 and
// sqrt(bc2) are formed by the host in double precision, as torch forms them (1.0f - 0.9f is not float(0.1)).
__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
       float step_size /* lr / bc1 */, float omb1 /* 1 - beta1 */, float beta2, float omb2 /* 1 - beta2 */, float eps, float wd,
       float bc2_sqrt) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            f32x4 pp = *reinterpret_cast<f32x4 *>(p + i), mm = *reinterpret_cast<f32x4 *>(m + i),
                  vv = *reinterpret_cast<f32x4 *>(v + i);
            const f32x4 gg0 = *reinterpret_cast<const f32x4 *>(g + i);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gg = gg0[c] + wd * pp[c];
                mm[c] = mm[c] + omb1 * (gg - mm[c]);
                vv[c] = vv[c] * beta2 + (omb2 * gg) * gg;
                const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
                pp[c] = pp[c] - step_size * (mm[c] / denom);
            }
            *reinterpret_cast<f32x4 *>(p + i) = pp;
            *reinterpret_cast<f32x4 *>(m + i) = mm;
            *reinterpret_cast<f32x4 *>(v + i) = vv;
        } else {
            for (long q = i; q < n; ++q) {
                const float gg = g[q] + wd * p[q];
                m[q] = m[q] + omb1 * (gg - m[q]);
                v[q] = v[q] * beta2 + (omb2 * gg) * gg;
                p[q] = p[q] - step_size * (m[q] / (sqrtf(v[q]) / bc2_sqrt + eps));
            }
        }
    }
}

// ------------------------------------------------------------- small pieces
// head[4] = {sum num_part, sum den_part, sum chosen_n, Bf}: the sums mcq_loss_tail consumes (one workgroup; each sum
// is 256 strided partials added in thread order, then a fixed tree)
__device__ __forceinline__ void loss_head_body(const float *__restrict__ num_part, const float *__restrict__ den_part, long nparts,
                                               const float *__restrict__ chosen_n, int N, float Bf, float *__restrict__ head,
                                               float (&s)[3][256]) {
    const int t = threadIdx.x;
    float a = 0.f, b = 0.f, c = 0.f;
    for (long i = t; i < nparts; i += 256) { a += num_part[i]; b += den_part[i]; }
    for (int i = t; i < N; i += 256) c += chosen_n[i];
    s[0][t] = a; s[1][t] = b; s[2][t] = c;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (t < m) { s[0][t] += s[0][t + m]; s[1][t] += s[1][t + m]; s[2][t] += s[2][t + m]; }
        __syncthreads();
    }
    if (t == 0) { head[0] = s[0][0]; head[1] = s[1][0]; head[2] = s[2][0]; head[3] = Bf; }
}

__global__ void __launch_bounds__(256)
k_loss_head(const float *__restrict__ num_part, const float *__restrict__ den_part, long nparts,
            const float *__restrict__ chosen_n, int N, float Bf, float *__restrict__ head) {
    __shared__ float s[3][256];
    loss_head_body(num_part, den_part, nparts, chosen_n, N, Bf, head, s);
}

// k_loss_head and k_loss_tail in one launch (a single process: no all-reduce of the sums between them); the sums pass
// through shared memory, `head` is written as well
__global__ void __launch_bounds__(256)
k_loss_head_tail(const float *__restrict__ num_part, const float *__restrict__ den_part, long nparts,
                 const float *__restrict__ chosen_n, int N, float Bf, float *__restrict__ head,
                 const float *__restrict__ prob_sum, const float *__restrict__ count, int K, float entropy_scale,
                 float *__restrict__ losses, float *__restrict__ g, float *__restrict__ g_prob) {
    __shared__ float s[3][256];
    loss_head_body(num_part, den_part, nparts, chosen_n, N, Bf, head, s);
    __syncthreads();
    const float num = s[0][0], den = s[1][0], chosen = s[2][0];
    __syncthreads();
    loss_tail_body(num, den, chosen, Bf, prob_sum, count, N, K, entropy_scale, losses, g, g_prob);
}

// The two scalar gradients from per-wave partials (fixed order: 1,024 strided sums, then a tree):
//   out_c = (sum part_c) * (sa[0] * sb[0] * sc) * speed     d/d centers_scale (:78: scaled centers = exp(speed*cs) * centers)
//   out_l = (sum part_l) * speed                            d/d logits_scale  (:278)
__global__ void __launch_bounds__(1024)
k_grad_tail(const float *__restrict__ part_c, long n_c, const float *__restrict__ sa, const float *__restrict__ sb, float sc,
            const float *__restrict__ part_l, long n_l, float speed, float *__restrict__ out_c, float *__restrict__ out_l) {
    __shared__ float s[2][1024];
    const int t = threadIdx.x;
    // thread t adds the elements t, t + 1024, ... in order, eight loads in flight at a time
    auto strided_sum = [&](const float *__restrict__ p, long n) {
        float a = 0.f;
        for (long i0 = t; i0 < n; i0 += 8 * 1024) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const long i = i0 + (long)u * 1024; v[u] = i < n ? p[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        return a;
    };
    s[0][t] = strided_sum(part_c, n_c);
    s[1][t] = strided_sum(part_l, n_l);
    __syncthreads();
    for (int m = 512; m >= 1; m >>= 1) {
        if (t < m) { s[0][t] += s[0][t + m]; s[1][t] += s[1][t + m]; }
        __syncthreads();
    }
    if (t == 0) {
        if (out_c) *out_c = s[0][0] * (((sa ? *sa : 1.0f) * (sb ? *sb : 1.0f) * sc) * speed);
        if (out_l) *out_l = s[1][0] * speed;
    }
}

// out[0] = exp(speed * centers_scale), out[1] = exp(speed * logits_scale)   (:78, :278; training flavour: on the device)
__global__ void k_scales(const float *__restrict__ centers_scale, const float *__restrict__ logits_scale, float speed,
                         float *__restrict__ out) {
    if (threadIdx.x == 0) out[0] = expf(*centers_scale * speed);
    if (threadIdx.x == 1) out[1] = expf(*logits_scale * speed);
}

// mean[d] = sum_n (sum_k C[n][k][d]) / K   (get_data_mean, :67-75: centers.mean(dim=1).sum(dim=0)).
// Workgroup = 16 columns x 16 waves; wave w takes the codebooks w, w + 16, ...; its lanes are 16 columns x 4 quarters of
// the entries: a quarter is added k ascending, the quarters as (q0 + q1) + (q2 + q3); wave 0 adds the codebook means, n
// ascending.
__global__ void __launch_bounds__(1024)
k_centers_mean(const float *__restrict__ C /*[N][K][Dp]*/, int N, int K, int Dp, float *__restrict__ mean /*[Dp]*/,
               float *__restrict__ cmean /*[N][Dp]: the codebooks' own means mu_n, or nullptr*/) {
    __shared__ float part[64][16];      // [codebook][column]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int d = blockIdx.x * 16 + c;          // Dp is a multiple of 16
    const int kn = K / 4;                       // K >= 16
    for (int n = wave; n < N; n += 16) {
        const float *p = C + ((size_t)n * K + (size_t)kq * kn) * Dp + d;
        float s = 0.f;
        for (int k0 = 0; k0 < kn; k0 += 4) {          // kn is a multiple of 4 (K >= 16): four loads in flight, added in order
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = p[(size_t)(k0 + u) * Dp];
#pragma unroll
            for (int u = 0; u < 4; ++u) s = s + r[u];
        }
        s = s + __shfl_xor(s, 16, 64);
        s = s + __shfl_xor(s, 32, 64);
        if (kq == 0) {
            part[n][c] = s / (float)K;
            if (cmean) cmean[(size_t)n * Dp + d] = s / (float)K;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = part[0][c];
        for (int n = 1; n < N; ++n) t = t + part[n][c];
        mean[d] = t;
    }
}

}  // namespace mcq
