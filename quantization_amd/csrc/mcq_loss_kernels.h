// mcq_loss_kernels.h -- HIP kernels (gfx950) for the loss of Quantizer.compute_loss
// (quantization/quantization.py:211-242) and its gradient: what QuantizerTrainer.step (:641-719)
// runs besides the index search.  All batch reductions are two-stage with a fixed order
// (no float atomics): results are bit-reproducible run to run.
//
//   logprobs  lp[b][n][k] = z[b][n][k] - lse[b][n]                    (log_softmax, :223)
//   chosen    Sum_b lp[b][n][idx[b][n]]                                (:225-227, before the mean)
//   prob_sum  Sum_b exp(lp[b][n][k])                                   (:238, before the mean)
//   count     #{b : idx[b][n] == k}                                    (:231-234, before the mean)
//   recon     err = decode(idx) - x,  Sum err^2,  Sum (x - mean)^2     (:213-217)
#pragma once
#include "mcq_kernels.h"

namespace mcq {

constexpr int kLossWaves = 4;   // waves per workgroup of the loss kernels

// Lanes of one logits row: KL = min(K, 64) lanes hold VPL = K / KL values each (k = lane_in_row + KL * i);
// a wave works on RPW = 64 / KL rows at a time.  Reductions stay inside the row's lane group.
template <int KL>
__device__ __forceinline__ float row_max(float v) {
#pragma unroll
    for (int m = KL / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
template <int KL>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int m = KL / 2; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

// Forward statistics.  Workgroup (n, chunk): kLossWaves waves share the rows b in
// [chunk * rows_per_chunk, +rows_per_chunk) of codebook n.  Outputs per (chunk, n): partial prob sums and
// counts [K], partial chosen sum; per row: lse.
template <int K>
__global__ void __launch_bounds__(64 * kLossWaves)
k_loss_fwd(const float *__restrict__ logits, const int64_t *__restrict__ idx, long B, int N, long rows_per_chunk,
           float *__restrict__ lse_out /*[B][N]*/, float *__restrict__ part_prob /*[chunks][N][K]*/,
           float *__restrict__ part_count /*[chunks][N][K]*/, float *__restrict__ part_chosen /*[chunks][N]*/) {
    constexpr int KL = K < 64 ? K : 64, VPL = K / KL, RPW = 64 / KL;
    __shared__ float s_prob[kLossWaves * RPW][K];
    __shared__ int s_count[K];
    __shared__ float s_chosen[kLossWaves];
    const int n = blockIdx.x % N;
    const long chunk = blockIdx.x / N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane / KL, kl = lane % KL;
    for (int k = tid; k < K; k += blockDim.x) s_count[k] = 0;
    __syncthreads();
    const long b_lo = chunk * rows_per_chunk;
    const long b_hi = (b_lo + rows_per_chunk < B) ? b_lo + rows_per_chunk : B;
    float acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc[i] = 0.f;
    float chosen = 0.f;
    // four rows per trip: their logits (and targets) are requested together, then taken in row order -- one row per trip
    // was a memory round trip per row (16 per wave at 64 rows per chunk)
    constexpr int UR = 4;
    for (long b0 = b_lo + (long)wave * RPW; b0 < b_hi; b0 += (long)UR * kLossWaves * RPW) {
        float vv[UR][VPL];
        int kiv[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const long b = b0 + (long)u * kLossWaves * RPW + sub;
            const long bc = b < b_hi ? b : b_hi - 1;
            const float *z = logits + (bc * N + n) * (long)K;
#pragma unroll
            for (int i = 0; i < VPL; ++i) vv[u][i] = z[kl + KL * i];
            kiv[u] = (int)idx[bc * N + n];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const long b = b0 + (long)u * kLossWaves * RPW + sub;
            const bool ok = b < b_hi;
            float (&v)[VPL] = vv[u];
            float mx = v[0];
#pragma unroll
            for (int i = 1; i < VPL; ++i) mx = fmaxf(mx, v[i]);
            mx = row_max<KL>(mx);
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) se += expf(v[i] - mx);
            se = row_sum<KL>(se);
            const float lse = mx + logf(se);
            const int ki = kiv[u];
            if (ok && kl == 0) lse_out[b * N + n] = lse;
            if (ok && ki >= 0) {   // negative target = "no index here" (padding frames of JointCodebookLoss): contributes nothing
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const float lp = v[i] - lse;
                    acc[i] += expf(lp);
                    if (kl + KL * i == ki) chosen += lp;
                }
                if (kl == 0) atomicAdd(&s_count[ki & (K - 1)], 1);   // integer: order-independent
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) s_prob[wave * RPW + sub][kl + KL * i] = acc[i];
    chosen = wave_sum_butterfly(chosen);
    if (lane == 0) s_chosen[wave] = chosen;
    __syncthreads();
    const long o = (chunk * N + n) * (long)K;
    for (int k = tid; k < K; k += blockDim.x) {
        float p = s_prob[0][k];
#pragma unroll
        for (int j = 1; j < kLossWaves * RPW; ++j) p += s_prob[j][k];
        part_prob[o + k] = p;
        part_count[o + k] = (float)s_count[k];
    }
    if (tid == 0) {
        float c = s_chosen[0];
#pragma unroll
        for (int w = 1; w < kLossWaves; ++w) c += s_chosen[w];
        part_chosen[chunk * N + n] = c;
    }
}

// Second stage: chunk partials -> sums, chunks ascending.  One workgroup per codebook.
__global__ void k_loss_reduce(const float *__restrict__ part_prob, const float *__restrict__ part_count,
                              const float *__restrict__ part_chosen, long chunks, int N, int K,
                              float *__restrict__ prob_sum /*[N][K]*/, float *__restrict__ count /*[N][K]*/,
                              float *__restrict__ chosen_sum /*[N]*/) {
    const int n = blockIdx.x;
    constexpr int U = 16;   // loads in flight; the additions stay in chunk order
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float p = 0.f, c = 0.f;
        for (long c0 = 0; c0 < chunks; c0 += U) {
            float tp[U], tc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long ch = (c0 + u < chunks) ? c0 + u : chunks - 1;
                tp[u] = part_prob[(ch * N + n) * (long)K + k];
                tc[u] = part_count[(ch * N + n) * (long)K + k];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c0 + u < chunks) { p += tp[u]; c += tc[u]; }
        }
        prob_sum[n * K + k] = p;
        count[n * K + k] = c;
    }
    if (threadIdx.x == blockDim.x - 1) {
        float s = 0.f;
        for (long c0 = 0; c0 < chunks; c0 += U) {
            float t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = part_chosen[((c0 + u < chunks) ? c0 + u : chunks - 1) * N + n];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c0 + u < chunks) s += t[u];
        }
        chosen_sum[n] = s;
    }
}

// Gradient w.r.t. the logits of  L = g_chosen * chosen + Sum_{n,k} g_prob[n][k] * prob_sum[n][k]:
//   dL/dz[b][n][k] = g_chosen * (delta(k, idx) - p) + p * (g_prob[n][k] - Sum_j p_j g_prob[n][j]),  p = exp(z - lse)
template <int K>
__global__ void __launch_bounds__(64 * kLossWaves)
k_loss_bwd(const float *__restrict__ logits, const int64_t *__restrict__ idx, const float *__restrict__ lse_in,
           long B, int N, const float *__restrict__ g_chosen /*[1]*/, const float *__restrict__ g_prob /*[N][K]*/,
           float *__restrict__ grad /*[B][N][K]*/, const float *__restrict__ bias = nullptr /*[N][K]*/,
           float *__restrict__ dot_part = nullptr /*[waves]: sum grad * (z - bias), for d/d logits_scale*/) {
    constexpr int KL = K < 64 ? K : 64, VPL = K / KL, RPW = 64 / KL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / KL, kl = lane % KL;
    const long row = ((long)blockIdx.x * kLossWaves + wave) * RPW + sub;   // row = b * N + n
    const bool ok = row < B * N;
    const long rc = ok ? row : B * N - 1;
    const int n = (int)(rc % N);
    const float gc = *g_chosen;
    const float lse = lse_in[rc];
    const int ki = (int)idx[rc];
    const float *z = logits + rc * (long)K;
    float p[VPL], g[VPL], dot = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        p[i] = expf(z[kl + KL * i] - lse);
        g[i] = g_prob[n * K + kl + KL * i];
        dot += p[i] * g[i];
    }
    dot = row_sum<KL>(dot);
    float ls = 0.f;
    if (ok) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float d = (kl + KL * i == ki) ? 1.f : 0.f;
            const float gv = (ki < 0) ? 0.f : gc * (d - p[i]) + p[i] * (g[i] - dot);
            grad[rc * (long)K + kl + KL * i] = gv;
            if (dot_part != nullptr) ls += gv * (z[kl + KL * i] - bias[n * K + kl + KL * i]);
        }
    }
    if (dot_part != nullptr) {
        ls = wave_sum_butterfly(ls);
        if (lane == 0) dot_part[(long)blockIdx.x * kLossWaves + wave] = ls;
    }
}

// Reconstruction pieces (:213-217): one wave per vector.  err[b] = (Sum_n C[n][idx[b][n]], n ascending) - x[b];
// per-workgroup partial sums of err^2 and (x - mean)^2 (4 vectors each), summed by the caller in order.
__global__ void __launch_bounds__(256)
k_recon_fwd(const float *__restrict__ x, const int64_t *__restrict__ idx, long B, const float *__restrict__ C,
            const float *__restrict__ mean /*[D]*/, int N, int K, int D, int Dp, float *__restrict__ err /*[B][D]*/,
            float *__restrict__ num_part, float *__restrict__ den_part) {
    __shared__ float s_num[4], s_den[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long b = (long)blockIdx.x * 4 + wave;
    float pn = 0.f, pd = 0.f;
    if (b < B) {
        const int64_t *id = idx + b * N;
        const bool vec = ((D & 3) == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(err) |
                                               reinterpret_cast<uintptr_t>(mean)) & 15) == 0);
        if (vec) {                       // 16 bytes per lane and row piece: a quarter of the load instructions
            for (int q = lane; q < D / 4; q += 64) {
                f32x4 t = *reinterpret_cast<const f32x4 *>(C + ((long)0 * K + (int)(id[0] & (K - 1))) * Dp + 4 * q);
                for (int n = 1; n < N; ++n)
                    t = t + *reinterpret_cast<const f32x4 *>(C + ((long)n * K + (int)(id[n] & (K - 1))) * Dp + 4 * q);
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + b * D + 4 * q);
                const f32x4 e = t - xv;
                *reinterpret_cast<f32x4 *>(err + b * D + 4 * q) = e;
                const f32x4 c = xv - *reinterpret_cast<const f32x4 *>(mean + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) { pn = fmaf(e[i], e[i], pn); pd = fmaf(c[i], c[i], pd); }
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                float t = C[((long)0 * K + (int)(id[0] & (K - 1))) * Dp + d];
                for (int n = 1; n < N; ++n) t = t + C[((long)n * K + (int)(id[n] & (K - 1))) * Dp + d];
                const float xv = x[b * D + d];
                const float e = t - xv;
                err[b * D + d] = e;
                pn = fmaf(e, e, pn);
                const float c = xv - mean[d];
                pd = fmaf(c, c, pd);
            }
        }
    }
    pn = wave_sum_butterfly(pn);
    pd = wave_sum_butterfly(pd);
    if (lane == 0) { s_num[wave] = pn; s_den[wave] = pd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        num_part[blockIdx.x] = ((s_num[0] + s_num[1]) + s_num[2]) + s_num[3];
        den_part[blockIdx.x] = ((s_den[0] + s_den[1]) + s_den[2]) + s_den[3];
    }
}

// The (N, K)-sized arithmetic of compute_loss (:217-241) and of QuantizerTrainer.step's total loss (:682-683)
// from the batch sums, plus the upstream gradients the backward kernels need.  One workgroup.
//   sums = {num, den, chosen (sum over codebooks), Btot}   (device floats: in data-parallel training the all-reduced ones)
//   losses[0..3] = rel_reconstruction, logprob, logits_entropy, index_entropy
//   total = rel + logprob + entropy_scale * logits_entropy:
//   g[0] = d total / d num = 1 / (den + 1e-20);  g[1] = d total / d chosen = -1 / (Btot * N);
//   g_prob[n][k] = d total / d prob_sum[n][k] = entropy_scale * (log(pbar) + 1) / (ref * N * Btot)
__device__ __forceinline__ void loss_tail_body(float num, float den, float chosen, float Bt, const float *__restrict__ prob_sum,
                                               const float *__restrict__ count, int N, int K, float entropy_scale,
                                               float *__restrict__ losses, float *__restrict__ g, float *__restrict__ g_prob) {
    // wave w takes the codebooks w, w + 4, ...: a codebook's two entropies are 64 lane-partial sums (entries k = lane,
    // lane + 64, ... ascending) and one xor butterfly -- no workgroup barrier inside the loop (the first version reduced
    // each codebook through a 256-thread LDS tree, nine barriers per codebook: 17.5 us at 16 codebooks)
    __shared__ float s_hl[64], s_hi[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float ref = logf((float)K);
    const float gscale = entropy_scale / (ref * (float)N * Bt);
    for (int n = wave; n < N; n += 4) {
        float a = 0.f, b = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float p = prob_sum[n * K + k] / Bt + 1.0e-20f;
            const float lp = logf(p);
            a = a + p * lp;
            g_prob[n * K + k] = (lp + 1.0f) * gscale;
            const float c = count[n * K + k] / Bt + 1.0e-20f;
            b = b + c * logf(c);
        }
        a = wave_sum_butterfly(a);
        b = wave_sum_butterfly(b);
        if (lane == 0) { s_hl[n] = -a; s_hi[n] = -b; }
    }
    __syncthreads();
    if (tid == 0) {
        float h_logits = 0.f, h_index = 0.f;     // sums over n ascending of the per-codebook entropies
        for (int n = 0; n < N; ++n) { h_logits += s_hl[n]; h_index += s_hi[n]; }
        losses[0] = num / (den + 1.0e-20f);
        losses[1] = -chosen / (Bt * (float)N);
        losses[2] = (ref - h_logits / (float)N) / ref;
        losses[3] = (ref - h_index / (float)N) / ref;
        g[0] = 1.0f / (den + 1.0e-20f);
        g[1] = -1.0f / (Bt * (float)N);
    }
}

__global__ void __launch_bounds__(256)
k_loss_tail(const float *__restrict__ sums, const float *__restrict__ prob_sum, const float *__restrict__ count, int N,
            int K, float entropy_scale, float *__restrict__ losses, float *__restrict__ g, float *__restrict__ g_prob) {
    loss_tail_body(sums[0], sums[1], sums[2], sums[3], prob_sum, count, N, K, entropy_scale, losses, g, g_prob);
}

// ------------------------------------------------------------ JointCodebookLoss
// quantization/prediction.py:38-66: the hidden activations that predict codebook n from the predictor and the
// entries chosen in codebooks 0..n-1.  One wave per frame b:
//   s_0 = hp[b];  s_n = s_{n-1} + scale * emb[(n-1) * K + max(idx[b][n-1], 0)];  A[n][b] = relu(s_n)
// (embedding * scale, cat, cumsum over the codebook axis, relu -- in that operation order).  A is laid out
// [N][B][H]: the per-codebook GEMMs that follow read contiguous batches.
__global__ void __launch_bounds__(256)
k_jcl_prefix_fwd(const float *__restrict__ hp /*[B][H]*/, const float *__restrict__ emb /*[(N-1)*K][H]*/,
                 const int64_t *__restrict__ idx /*[B][N]*/, long B, int N, int K, int H, float scale,
                 float *__restrict__ A /*[N][B][H]*/) {
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    for (int h = lane; h < H; h += 64) {
        float s = hp[b * H + h];
        A[b * H + h] = fmaxf(s, 0.f);
        for (int n = 1; n < N; ++n) {
            long k = idx[b * N + n - 1];
            k = k < 0 ? 0 : k;
            s = s + emb[((long)(n - 1) * K + k) * H + h] * scale;
            A[((long)n * B + b) * H + h] = fmaxf(s, 0.f);
        }
    }
}

// Backward of the above given gA = dL/dA: with m_n = (A[n][b] > 0) ? gA[n][b] : 0 and the reverse running sum
// r_n = m_n + r_{n+1}:  d hp[b] = r_0;  gE[n-1][b] = scale * r_n is the gradient of the embedding row chosen by
// frame b for codebook n-1 (scattered to the table by mcq_scatter_rows).
__global__ void __launch_bounds__(256)
k_jcl_prefix_bwd(const float *__restrict__ A, const float *__restrict__ gA, long B, int N, int H, float scale,
                 float *__restrict__ g_hp /*[B][H]*/, float *__restrict__ gE /*[N-1][B][H]*/) {
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    for (int h = lane; h < H; h += 64) {
        float r = 0.f;
        for (int n = N - 1; n >= 1; --n) {
            const long o = ((long)n * B + b) * H + h;
            r = r + ((A[o] > 0.f) ? gA[o] : 0.f);
            gE[((long)(n - 1) * B + b) * H + h] = r * scale;
        }
        r = r + ((A[b * H + h] > 0.f) ? gA[b * H + h] : 0.f);
        g_hp[b * H + h] = r;
    }
}

// uint8 working indexes -> int64 (argmax output of mcq_logits_argmax)
__global__ void k_export_indexes(const uint8_t *__restrict__ in, long n, int64_t *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

}  // namespace mcq
