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
