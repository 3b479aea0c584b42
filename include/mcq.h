/*
 * mcq.h -- C ABI of libmcq_hip.so: the MI355X (gfx950) index search and decode
 * of the multi-codebook quantizer.
 *
 * The reference (danpovey/quantization) has no FFI for this path: it is a Python
 * class whose private methods call torch ops.  Each entry point below replaces
 * the named reference method; the host-side mirror (quantization_amd/quantizer.py)
 * binds them with ctypes -- INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - every pointer except `prepared`/`workspace` sizes is a BORROWED DEVICE pointer
 *    (torch `tensor.data_ptr()`), contiguous, alive until the stream has drained;
 *  - every function only enqueues work on `stream` (a hipStream_t passed as void*;
 *    NULL = the default stream) and returns immediately: no allocation, no
 *    synchronisation, re-entrant per (device, stream).  Process-wide state is limited to
 *    read-only tuning hooks (environment variables latched on first use; DESIGN.md lists
 *    them; none changes a result) and a thread-local launch counter
 *    (mcq_last_encode_launches);
 *  - return value: 0 = ok; MCQ_E* < 0 = rejected argument; > 0 = hipError_t of a
 *    failed launch.  Nothing is thrown across the boundary.
 *  - supported domain: codebook_size K a power of two in [16, 1024], num_codebooks N
 *    a power of two, N <= 64, N*K <= 16384 (the reference's trainer produces at most 64 x 16 and 32 x 256:
 *    bytes_per_frame <= 32, quantization/quantization.py:614; `prepared` holds the N*K x N*K Gram matrix).
 *    K = 512 / 1024 (Quantizer(codebook_size=...) with as_bytes=False, :35): the index search, mcq_refine_indexes,
 *    mcq_logits and mcq_decode (int64 codes); every uint8 output must be NULL there (MCQ_EINVAL otherwise), and the
 *    trainer's entry points (mcq_logits_argmax, mcq_logits_refine, mcq_logits_refine_codes, mcq_loss_*, mcq_recon_fwd,
 *    mcq_decode_backward_u8(_ex), ...) answer MCQ_EUNSUPPORTED.  Any
 *    dim 1 <= D <= 16384 (rows are zero-padded to a multiple of 16 inside `prepared`; the i32 accumulators
 *    of the fixed-point products bound D).  The reference crashes for K < 16
 *    (quantization/quantization.py:506) and needs K <= 256 for byte output (:271).
 *
 * Numerics: bit-identical to oracle/mcq_oracle.c (see its header for the spec).  The
 * three inner-product tables of the path -- the logits, x.C and the Gram matrix of the
 * centers -- are EXACT fixed-point products ("fixdot": rows as 30-bit fixed point against
 * their own largest magnitude, ten 8-bit limb products on the i8 matrix cores, one defined
 * fp32 combination): nothing in them depends on a summation order.  Sums of squares are
 * wave64 butterfly reductions; the refinement passes read their inner products from the
 * Gram matrix kept in `prepared` and from one x.C product per call (the TABLE FORM).
 * Non-finite inputs are outside the contract (the reference returns arbitrary codes for them).
 */
#ifndef MCQ_H
#define MCQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCQ_EINVAL (-1)     /* bad shape / null pointer                         */
#define MCQ_EUNSUPPORTED (-2) /* outside the supported (K, N) domain             */
#define MCQ_EWORKSPACE (-3) /* workspace smaller than mcq_encode_workspace_bytes */

#define MCQ_ABI_VERSION 7   /* 7: no signature changed.  The shortlists of a refinement pass are SETS listed in ascending position (they were
                               listed by (value, position) until 6: oracle/mcq_oracle.c::select_smallest) -- results differ from ABI 6 only
                               where two later scores tie exactly; mcq_test_select hands the list over in that order and takes the layout
                               in the sign of per_lane; 16 codebooks of 16 entries run all passes of a call in one launch (k_tf_pass16);
                               6: the products of the path are formed from CENTERED rows and frames (`prepared` also holds the codebooks' own
                               means and the classifier rows' products with the data mean, Q and the Gram matrix are those of
                               C[n][k] - mu_n, the frame planes of the workspace those of x - mean: sizes changed, signatures did
                               not), mcq_profile_encode times the shipped launch sequence and
                               reports per-category launch counts, mcq_profile_category_name is new;
                               5: mcq_prepared_decode_bytes, 64 codebooks for every codebook size, and (additions) mcq_prepare_params,
                               mcq_logits_refine_codes, mcq_loss_head_tail; 4: fixed-point products: `prepared` holds limb planes of the centers and the classifier
                               (mcq_prepared_bytes changed), workspaces hold those of the frames (the workspace sizes
                               depend on D), mcq_logits takes a workspace, mcq_logits_workspace_bytes is new */
int mcq_abi_version(void);

/* D rounded up to the padded row length used inside `prepared` and workspaces. */
int mcq_padded_dim(int D);

/* ---- derived state -------------------------------------------------------
 * Replaces Quantizer.get_centers() (quantization/quantization.py:77-79, recomputed
 * on every call there) and the parameter reads of Quantizer._logits (:277-279).
 * `prepared` receives: scaled centers C[N][K][Dp] = cscale_exp * centers (what decode sums), the codebooks' own means
 * mu_n = mean_k C[n][k] and their sum (= get_data_mean(), :67-75), and -- when weight is given -- what the search reads:
 * the CENTERED rows C[n][k] - mu_n as 8-bit limb planes with their row exponents, their sums of squares Q[N][K] (:411),
 * the rows of to_logits.weight as limb planes and their products with the data mean (the frames of both products are centered: a
 * logit is ((fixdot(x - mean, W_r) + fixdot(mean, W_r)) * lscale) + bias_r), the bias, and the Gram matrix G[N*K][N*K] of the
 * centered rows (16 MB at 8 x 256; what the refinement passes read).  The search is invariant under this shift
 * (oracle/mcq_oracle.c, "CENTERING").
 * cscale_exp / lscale_exp = exp(10*centers_scale) / exp(10*logits_scale), formed
 * by the caller in fp32 exactly as the reference does (:78, :278).
 * weight/bias may be NULL when only decode is needed: `prepared` then receives the scaled centers
 * only and needs mcq_prepared_decode_bytes (no limb planes, no Gram matrix: 16 MB .. 1 GB less);
 * such a state serves mcq_decode and nothing else.                                              */
size_t mcq_prepared_bytes(int N, int K, int D);
size_t mcq_prepared_decode_bytes(int N, int K, int D);
/* byte offset inside `prepared` of float[mcq_padded_dim(D)]: get_data_mean() (:67-75) of the scaled centers,
 * sum_n mean_k C[n][k][:] (the scaled centers themselves are at offset 0, [N][K][mcq_padded_dim(D)])            */
size_t mcq_prepared_mean_offset(int N, int K, int D);
int mcq_prepare(const float *centers, float cscale_exp, const float *weight, const float *bias,
                int N, int K, int D, void *prepared, void *stream);

/* As mcq_prepare with the two scale factors read from DEVICE memory: scales_exp = float[2]
 * {exp(10*centers_scale), exp(10*logits_scale)}.  No host copy of the (trained) scale parameters is
 * needed, so a training loop never synchronises; the logits factor is kept inside `prepared` and
 * used by mcq_encode_ex when MCQ_ENCODE_LSCALE_FROM_PREPARED is set.                            */
int mcq_prepare_dev(const float *centers, const float *scales_exp, const float *weight, const float *bias,
                    int N, int K, int D, void *prepared, void *stream);

/* As mcq_prepare_dev with the scale PARAMETERS read from device memory: centers_scale / logits_scale are the two scalar
 * parameters themselves, speed = 10 (Quantizer.scale_speed); exp(speed * scale) is formed on the device (the expf of
 * mcq_scales_exp) in the first kernel of the chain.  scales_exp_out (may be NULL) receives float[2]
 * {exp(speed*centers_scale), exp(speed*logits_scale)}: what the backward kernels take as `sa` / `scale_dev`.      */
int mcq_prepare_params(const float *centers, const float *centers_scale, const float *logits_scale, float speed,
                       const float *weight, const float *bias, int N, int K, int D, void *prepared,
                       float *scales_exp_out, void *stream);

/* ---- index search ----------------------------------------------------------
 * Replaces Quantizer._compute_indexes (:281-305): learned-logit argmax followed
 * by `refine_iters` passes of Quantizer._refine_indexes (:308-547).
 * x: fp32 [B][D].  Exactly one of out_u8 / out_i64 is non-NULL:
 *   out_i64: int64 [B][N]            -- _compute_indexes / encode(as_bytes=False)
 *   out_u8 : uint8 [B][N / pack]     -- encode(as_bytes=True) (:266-272), where
 *            pack = 2 when K == 16 (low nibble = even codebook), else 1.
 * workspace: device scratch of at least mcq_encode_workspace_bytes(B, N, K, D)
 * bytes (the batch is processed in chunks that fit; any B >= 0 is accepted).   */
size_t mcq_encode_workspace_bytes(long B, int N, int K, int D);
int mcq_encode(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
               int refine_iters, uint8_t *out_u8, int64_t *out_i64, void *workspace,
               size_t workspace_bytes, void *stream);

/* mcq_encode with options.  MCQ_ENCODE_SKIP_FIXED_POINTS: _refine_indexes is a deterministic map
 * of (x, indexes), so a vector whose indexes a pass leaves unchanged is already final; with this
 * flag such vectors leave the active list and later passes only process the rest.  The codes are
 * identical to mcq_encode's for every input; the cost becomes data dependent (off by default, and
 * never used for the headline benchmark figure).                                                */
#define MCQ_ENCODE_SKIP_FIXED_POINTS 1u
#define MCQ_ENCODE_LSCALE_FROM_PREPARED 2u /* lscale_exp argument ignored: see mcq_prepare_dev */
/* x points to IEEE fp16 [B][D] (the reference's data helper yields fp16 frames that callers widen,
 * quantization/quantization.py:798): rows widen to fp32 in the kernels' load path.  Every fp16 value
 * is an fp32 value, so the codes equal those of the widened input bit for bit.                   */
#define MCQ_ENCODE_X_FP16 4u
int mcq_encode_ex(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                  int refine_iters, uint8_t *out_u8, int64_t *out_i64, void *workspace,
                  size_t workspace_bytes, void *stream, unsigned flags);

/* Replaces Quantizer._refine_indexes (:308-547) applied `refine_iters` times to caller-supplied
 * indexes: idx_in int64 [B][N] with entries in [0, K) -> idx_out int64 [B][N] (may alias idx_in).
 * Same workspace as mcq_encode.                                                               */
int mcq_refine_indexes(const float *x, long B, const void *prepared, int N, int K, int D, int refine_iters,
                       const int64_t *idx_in, int64_t *idx_out, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---- decode ----------------------------------------------------------------
 * Replaces Quantizer.decode + _maybe_separate_indexes (:117-148, :551-573).
 * codes: [B][codes_per_row] of uint8 (code_bytes == 1) or int64 (code_bytes == 8);
 * codes_per_row == N, or N / r with r in {2,4,8,16} for packed codes (each code
 * holds r base-K digits, least significant first).  out: fp32 [B][D],
 * out[b] = sum over n ascending of C[n][index(b, n)].                          */
int mcq_decode(const void *codes, int code_bytes, int codes_per_row, long B, const void *prepared,
               int N, int K, int D, float *out, void *stream);

/* Gradient of decode w.r.t. the scaled centers (what autograd derives from the gather + sum of
 * :142-147): gC[n][k][:] = sum over the vectors b with index(b, n) == k, b ascending, of
 * grad_out[b][:].  Deterministic (fixed summation order, no atomics).  idx: int64 [B][N];
 * grad_out: fp32 [B][D]; gC: fp32 [N][K][D], fully overwritten.                                */
int mcq_decode_backward(const float *grad_out, const int64_t *idx, long B, int N, int K, int D, float *gC,
                        void *stream);

/* ---- trainer pieces ------------------------------------------------------------
 * What QuantizerTrainer.step (:641-719) runs besides the index search: the loss of
 * Quantizer.compute_loss (:211-242) as batch SUMS (the caller forms the means and ratios, and in
 * data-parallel training all-reduces the sums first) and its gradient.  Every reduction has a
 * fixed order: results are bit-reproducible.
 *
 * mcq_logits_argmax: logits fp32 [B][N*K] of Quantizer._logits (:277-279) AND their per-codebook
 * first-maximum argmax int64 [B][N] (:301) from one GEMM; the indexes then go through
 * mcq_refine_indexes.  workspace >= mcq_logits_workspace_bytes(B, N, D).  flags: MCQ_ENCODE_LSCALE_FROM_PREPARED, MCQ_ENCODE_X_FP16. */
int mcq_logits_argmax(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                      float *logits_out, int64_t *argmax_out, void *workspace, size_t workspace_bytes,
                      void *stream, unsigned flags);

/* mcq_logits_argmax followed by mcq_refine_indexes in one call (what compute_loss needs, :211-219): logits_out as above,
 * idx_out int64 [B][N] = the indexes after `refine_iters` passes.  workspace as for mcq_encode.             */
int mcq_logits_refine(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                      int refine_iters, float *logits_out, int64_t *idx_out, void *workspace, size_t workspace_bytes,
                      void *stream, unsigned flags);
/* the same, and the indexes a second time as unpacked bytes codes_out uint8 [B][N] (may be NULL): what
 * mcq_decode_backward_u8(_ex) scans -- the trainer's step needs both and would otherwise convert one into the other */
int mcq_logits_refine_codes(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                            int refine_iters, float *logits_out, int64_t *idx_out, uint8_t *codes_out, void *workspace,
                            size_t workspace_bytes, void *stream, unsigned flags);

/* Log-softmax statistics of logits [B][N*K] against indexes int64 [B][N] (:221-240):
 *   lse[b][n] = logsumexp_k;  chosen_sum[n] = sum_b (logit[b][n][idx] - lse);
 *   prob_sum[n][k] = sum_b softmax;  count[n][k] = #{b: idx[b][n] == k}.                       */
size_t mcq_loss_workspace_bytes(long B, int N, int K);
int mcq_loss_fwd(const float *logits, const int64_t *idx, long B, int N, int K, float *lse, float *chosen_sum,
                 float *prob_sum, float *count, void *workspace, size_t workspace_bytes, void *stream);

/* Gradient w.r.t. the logits of  g_chosen * sum_n chosen_sum[n] + sum_{n,k} g_prob[n][k] * prob_sum[n][k];
 * g_chosen (float[1]) and g_prob (float[N][K]) are DEVICE pointers (no host copy of upstream gradients). */
int mcq_loss_bwd(const float *logits, const int64_t *idx, const float *lse, long B, int N, int K,
                 const float *g_chosen, const float *g_prob, float *grad_logits, void *stream);

/* The (N, K)-sized tail of compute_loss (:217-241) and of the trainer's total loss
 * rel + logprob + entropy_scale * logits_entropy (:682-683), on DEVICE floats:
 *   sums = {sum err^2, sum (x-mean)^2, sum_n chosen_sum[n], total batch size}  (all-reduced in DP training)
 *   losses[4] = rel_reconstruction, logprob, logits_entropy, index_entropy losses;
 *   g[2] = d total / d sums[0], d total / d sums[2];  g_prob[N][K] = d total / d prob_sum
 * (g[1] and g_prob feed mcq_loss_bwd; 2 * g[0] scales mcq_decode_backward(err)).                */
int mcq_loss_tail(const float *sums, const float *prob_sum, const float *count, int N, int K, float entropy_scale,
                  float *losses, float *g, float *g_prob, void *stream);

/* Reconstruction pieces (:213-217): err[b] = decode(idx[b]) - x[b] (fp32 [B][D]); partial sums over
 * groups of 4 vectors of err^2 (num_part) and (x - mean)^2 (den_part), float[(B + 3) / 4] each, to be
 * summed by the caller; mean = get_data_mean() (:67-75), float[D].  d(sum err^2)/d(centers) is
 * 2 * mcq_decode_backward(err).                                                                   */
int mcq_recon_fwd(const float *x, const int64_t *idx, long B, const void *prepared, const float *mean, int N,
                  int K, int D, float *err, float *num_part, float *den_part, void *stream);


/* ---- parameter update of QuantizerTrainer.step (:708-715, :722-730) ------------------------
 * mcq_weight_grad: the autograd of Quantizer._logits (:277-279) w.r.t. to_logits: with G = dL/dlogits fp32 [B][M]
 *   (M = N*K, from mcq_loss_bwd) and the frames x fp32 [B][D]:  gW[M][D] = s * G^T x,  gb[M] = column sums of G,
 *   s = exp(10*logits_scale) read from DEVICE memory (scale_dev).  Splits of the batch whose partial tiles (workspace) are
 *   added in a fixed order; fp32 MFMA, or -- M and D multiples of 128, M >= 1024, B >= 2048 -- six bf16 piece products per
 *   multiply-add (each operand as three bf16 pieces: fp32-grade, 1e-6 of the largest entry against the fp64 product).
 *   Deterministic; not part of the bit-exact contract of the index search.
 * mcq_adam_step: torch.optim.Adam's update (weight decay as L2 term, no amsgrad) on one flat bucket of n floats:
 *   parameters p, gradients g, moments m / v; bias_correction1 = 1 - beta1^t and sqrt(1 - beta2^t) formed by the caller.
 * mcq_loss_head: head[4] = {sum num_part, sum den_part, sum chosen_n, batch}: mcq_loss_tail's `sums` from the
 *   partials of mcq_recon_fwd / mcq_loss_fwd without host or library reductions.
 * mcq_scales_exp: out2 = {exp(speed * centers_scale), exp(speed * logits_scale)} on the device (the `scales_exp`
 *   of mcq_prepare_dev).                                                                          */
size_t mcq_weight_grad_workspace_bytes(long B, int M, int D);   /* partial tiles of the batch splits */
int mcq_weight_grad(const float *G, const float *x, long B, int M, int D, const float *scale_dev, float *gW, float *gb,
                    void *workspace, size_t workspace_bytes, void *stream);
int mcq_adam_step(float *p, const float *g, float *m, float *v, long n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double bias_correction1, double bias_correction2_sqrt, void *stream);
int mcq_loss_head(const float *num_part, const float *den_part, long nparts, const float *chosen_n, int N, float batch,
                  float *head, void *stream);
int mcq_scales_exp(const float *centers_scale, const float *logits_scale, float speed, float *out2, void *stream);
/* mcq_loss_head followed by mcq_loss_tail in ONE launch (same results; for a single process, where no all-reduce of the
 * sums sits between the two) */
int mcq_loss_head_tail(const float *num_part, const float *den_part, long nparts, const float *chosen_n, int N, float batch,
                       float *head, const float *prob_sum, const float *count, int K, float entropy_scale, float *losses,
                       float *g, float *g_prob, void *stream);

/* The scalar gradients without library reductions.  mcq_decode_backward_u8_ex: mcq_decode_backward_u8 whose stored
 * rows are scaled by sa[0]*sb[0]*sc (device floats sa, sb; host float sc) and which also leaves, per wave, the share of
 * <unscaled sums, dotw> (dotw fp32 [N][K][D]) in dot_part[mcq_decode_backward_waves(N, K, D)].  mcq_loss_bwd_ex:
 * mcq_loss_bwd that also leaves sum grad * (logit - bias) per wave in dot_part[mcq_loss_bwd_waves(B, N, K)].
 * mcq_grad_tail reduces both partial arrays in a fixed order:
 *   out_c = (sum part_c) * sa[0]*sb[0]*sc * speed   (d/d centers_scale),   out_l = (sum part_l) * speed   (d/d logits_scale). */
long mcq_decode_backward_waves(int N, int K, int D);
int mcq_decode_backward_u8_ex(const float *grad_out, const uint8_t *codes, long B, int N, int K, int D, float *gC,
                              const float *sa, const float *sb, float sc, const float *dotw, float *dot_part, void *stream);
long mcq_loss_bwd_waves(long B, int N, int K);
int mcq_loss_bwd_ex(const float *logits, const int64_t *idx, const float *lse, long B, int N, int K, const float *g_chosen,
                    const float *g_prob, float *grad_logits, const float *bias, float *dot_part, void *stream);
int mcq_grad_tail(const float *part_c, long n_c, const float *sa, const float *sb, float sc, const float *part_l, long n_l,
                  float speed, float *out_c, float *out_l, void *stream);

/* ---- JointCodebookLoss pieces (quantization/prediction.py:9-82) ----------------------
 * The consumer of the codes: a predictor trained to predict codebook n from its input and the entries of
 * codebooks 0..n-1.  The GEMMs are library calls on the caller's side; these are the fused non-GEMM parts.
 *
 * mcq_jcl_prefix_fwd (:38-66): A[n][b][:] = relu(hp[b] + scale * sum_{m<n} emb[m*K + max(idx[b][m], 0)]),
 *   summed in codebook order (embedding * scale, cat, cumsum, relu); hp fp32 [B][H] (the output of linear1),
 *   emb fp32 [(N-1)*K][H], idx int64 [B][N], A fp32 [N][B][H].
 * mcq_jcl_prefix_bwd: from gA = dL/dA: g_hp [B][H] and gE [N-1][B][H], the gradient of the embedding row frame b
 *   chose for codebook n (to be scattered with mcq_scatter_rows).
 * mcq_scatter_rows: out[n][k][:] = sum over b ascending with idx[b*idx_stride + n] == k of
 *   grad[b*stride_b + n*stride_n + :D] -- mcq_decode_backward with per-(b, n) gradients; negative indexes
 *   (padding frames) match nothing.  The cross-entropy itself is mcq_loss_fwd / mcq_loss_bwd on [N*B][K] logits,
 *   whose negative targets contribute nothing (ignore_index, :78-81).                                   */
int mcq_jcl_prefix_fwd(const float *hp, const float *emb, const int64_t *idx, long B, int N, int K, int H,
                       float scale, float *A, void *stream);
int mcq_jcl_prefix_bwd(const float *A, const float *gA, long B, int N, int H, float scale, float *g_hp, float *gE,
                       void *stream);
int mcq_scatter_rows(const float *grad, long stride_b, long stride_n, const int64_t *idx, int idx_stride, long B,
                     int N, int K, int D, float *out, void *stream);

/* mcq_decode_backward on unpacked uint8 codes [B][N] (K <= 256): same sums in the same order; the kernel is
 * bound by scanning the index column, which is 8x smaller this way (what the trainer's step uses).      */
int mcq_decode_backward_u8(const float *grad_out, const uint8_t *codes, long B, int N, int K, int D, float *gC,
                           void *stream);

/* ---- test / profiling hooks -------------------------------------------------
 * Logits of Quantizer._logits (:277-279) for a batch, fp32 [B][N*K]; used by the
 * parity tests to localise a divergence.                                       */
size_t mcq_logits_workspace_bytes(long B, int N, int D);   /* also what mcq_logits_argmax needs */
int mcq_logits(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
               float *out, void *workspace, size_t workspace_bytes, void *stream);

/* The wave-level selection of the search on its own: `cases` independent problems of 64 * per_lane scores each
 * (per_lane 1, 4 or 16: key i of lane l at position per_lane * l + i; -4 / -16: the slot-major layout, position 64 * i + l; -1004:
 * four keys per lane handled as positions in no particular order); out_v / out_p [cases][64] receive the cnt smallest by
 * (value, position) LISTED IN ASCENDING POSITION (a list that runs out of candidates is padded with (INF, M - 1)).
 * Test hook for the selection's paths (ties, clustered survivors).                                                       */
int mcq_test_select(const float *scores, int cases, int per_lane, int cnt, float *out_v, int *out_p, void *stream);

/* Name and launch count of the kernels enqueued by the last mcq_encode on this
 * thread (for bench.py's per-kernel HIP-event timing); returns the count.      */
int mcq_last_encode_launches(void);

/* Measurement tool (bench.py): runs the encode exactly as mcq_encode enqueues it -- the same launches, nothing switched off --
 * once per category of launch, with HIP events on `stream` round the launches of that category only (events round every launch of
 * one encode stretch it by a tenth); synchronises and allocates the output of those encodes.
 * ms_out[c] / launches_out[c] (c < cap) receive the summed milliseconds and the number of timed intervals of category c,
 * mcq_profile_category_name(c) its name (NULL past the last one).  Returns the number of categories, or an error code.   */
int mcq_profile_encode(const float *x, long B, const void *prepared, float lscale_exp, int N, int K,
                       int D, int refine_iters, void *workspace, size_t workspace_bytes, void *stream,
                       float *ms_out, int *launches_out, int cap);
const char *mcq_profile_category_name(int category);

#ifdef __cplusplus
}
#endif
#endif /* MCQ_H */
