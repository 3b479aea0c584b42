/*
 * mcq_host.c -- host twins of the C ABI (see mcq_host.h).  TEST INFRASTRUCTURE ONLY.
 * The host `prepared` blob keeps the caller's arguments (header, raw centers, weight, bias); every
 * call rebuilds the oracle's derived state from it, so nothing is allocated behind the caller's back
 * beyond the call's own scratch.  Argument checks mirror quantization_amd/csrc/mcq_api.hip.
 */
#include "mcq_host.h"

#include <stdlib.h>
#include <string.h>

#define MCQ_EINVAL (-1)
#define MCQ_EUNSUPPORTED (-2)

typedef struct mcq_oracle mcq_oracle;
mcq_oracle *mcq_oracle_create(const float *centers, float cscale_exp, const float *W, const float *bias,
                              float lscale_exp, int N, int K, int D);
void mcq_oracle_free(mcq_oracle *o);
typedef uint16_t mcq_code;      /* as in mcq_oracle.c */
int mcq_oracle_compute_indexes(const mcq_oracle *o, const float *x, long B, int iters, mcq_code *idx, int nthreads);
int mcq_oracle_refine(const mcq_oracle *o, const float *x, long B, int iters, mcq_code *idx, int nthreads);
int mcq_oracle_logits(const mcq_oracle *o, const float *x, long B, float *logits);
void mcq_oracle_decode(const mcq_oracle *o, const mcq_code *idx, long B, float *out);

typedef struct {
    uint32_t magic;
    int N, K, D, has_w;
    float cscale;
    uint32_t pad[2];
} host_hdr;
#define HOST_MAGIC 0x4d435148u

static int is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static int domain_ok(int N, int K, int D) {
    return is_pow2(K) && K >= 16 && K <= 1024 && is_pow2(N) && N <= 64 && (long)N * K <= 16384 && D >= 1 && D <= 16384;
}
static int domain_err(int N, int K) { return (K < 16 || K > 1024 || N > 64 || (long)N * K > 16384) ? MCQ_EUNSUPPORTED : MCQ_EINVAL; }

size_t mcq_prepared_bytes_host(int N, int K, int D) {
    size_t nk = (size_t)N * K;
    return sizeof(host_hdr) + sizeof(float) * (2 * nk * D + nk);
}

int mcq_prepare_host(const float *centers, float cscale_exp, const float *weight, const float *bias,
                     int N, int K, int D, void *prepared, void *stream) {
    (void)stream;
    if (!domain_ok(N, K, D)) return domain_err(N, K);
    if (!centers || !prepared || ((weight == NULL) != (bias == NULL))) return MCQ_EINVAL;
    size_t nk = (size_t)N * K;
    host_hdr h = {HOST_MAGIC, N, K, D, weight != NULL, cscale_exp, {0, 0}};
    memcpy(prepared, &h, sizeof(h));
    float *p = (float *)((char *)prepared + sizeof(h));
    memcpy(p, centers, sizeof(float) * nk * D);
    if (weight) {
        memcpy(p + nk * D, weight, sizeof(float) * nk * D);
        memcpy(p + 2 * nk * D, bias, sizeof(float) * nk);
    }
    return 0;
}

static mcq_oracle *open_state(const void *prepared, float lscale, int N, int K, int D, int need_w) {
    host_hdr h;
    memcpy(&h, prepared, sizeof(h));
    if (h.magic != HOST_MAGIC || h.N != N || h.K != K || h.D != D || (need_w && !h.has_w)) return NULL;
    size_t nk = (size_t)N * K;
    const float *p = (const float *)((const char *)prepared + sizeof(h));
    return mcq_oracle_create(p, h.cscale, h.has_w ? p + nk * D : NULL, h.has_w ? p + 2 * nk * D : NULL, lscale,
                             N, K, D);
}

size_t mcq_encode_workspace_bytes_host(long B, int N, int K, int D) {
    (void)B;
    return domain_ok(N, K, D) ? 1 : 0;
}

/* encode tail (quantization/quantization.py:266-275): nibble packing when K == 16, cast */
static void write_codes(const mcq_code *idx, long B, int N, int K, uint8_t *out_u8, int64_t *out_i64) {
    if (out_i64) {
        for (size_t i = 0; i < (size_t)B * N; i++) out_i64[i] = idx[i];
    } else if (K == 16 && N >= 2) {
        for (size_t i = 0; i < (size_t)B * N / 2; i++) out_u8[i] = (uint8_t)(idx[2 * i] | (idx[2 * i + 1] << 4));
    } else {
        for (size_t i = 0; i < (size_t)B * N; i++) out_u8[i] = (uint8_t)idx[i];
    }
}

int mcq_encode_host(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                    int refine_iters, uint8_t *out_u8, int64_t *out_i64, void *workspace,
                    size_t workspace_bytes, void *stream) {
    (void)stream; (void)workspace_bytes;
    if (!domain_ok(N, K, D)) return domain_err(N, K);
    if (B < 0 || refine_iters < 0 || refine_iters > 60 || (out_u8 == NULL) == (out_i64 == NULL)) return MCQ_EINVAL;
    if (K > 256 && out_u8 != NULL) return MCQ_EINVAL;      /* (:271: bytes hold entries of up to 256-entry codebooks) */
    if (B == 0) return 0;
    if (!x || !prepared || !workspace) return MCQ_EINVAL;
    mcq_oracle *o = open_state(prepared, lscale_exp, N, K, D, 1);
    if (!o) return MCQ_EINVAL;
    mcq_code *idx = (mcq_code *)malloc(sizeof(mcq_code) * (size_t)B * N);
    int rc = mcq_oracle_compute_indexes(o, x, B, refine_iters, idx, 0);
    if (rc == 0) write_codes(idx, B, N, K, out_u8, out_i64);
    free(idx);
    mcq_oracle_free(o);
    return rc;
}

int mcq_refine_indexes_host(const float *x, long B, const void *prepared, int N, int K, int D, int refine_iters,
                            const int64_t *idx_in, int64_t *idx_out, void *workspace, size_t workspace_bytes,
                            void *stream) {
    (void)stream; (void)workspace_bytes;
    if (!domain_ok(N, K, D)) return domain_err(N, K);
    if (B < 0 || refine_iters < 0 || refine_iters > 60) return MCQ_EINVAL;
    if (B > 0 && (!idx_in || !idx_out)) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!x || !prepared || !workspace) return MCQ_EINVAL;
    mcq_oracle *o = open_state(prepared, 1.0f, N, K, D, 0);
    if (!o) return MCQ_EINVAL;
    mcq_code *idx = (mcq_code *)malloc(sizeof(mcq_code) * (size_t)B * N);
    for (size_t i = 0; i < (size_t)B * N; i++) {
        if (idx_in[i] < 0 || idx_in[i] >= K) { free(idx); mcq_oracle_free(o); return MCQ_EINVAL; }
        idx[i] = (mcq_code)idx_in[i];
    }
    int rc = mcq_oracle_refine(o, x, B, refine_iters, idx, 0);
    if (rc == 0)
        for (size_t i = 0; i < (size_t)B * N; i++) idx_out[i] = idx[i];
    free(idx);
    mcq_oracle_free(o);
    return rc;
}

int mcq_decode_host(const void *codes, int code_bytes, int codes_per_row, long B, const void *prepared,
                    int N, int K, int D, float *out, void *stream) {
    (void)stream;
    if (!domain_ok(N, K, D)) return domain_err(N, K);
    if (B < 0 || codes_per_row <= 0 || N % codes_per_row != 0) return MCQ_EINVAL;
    const int rep = N / codes_per_row;
    if (!(rep == 1 || rep == 2 || rep == 4 || rep == 8 || rep == 16)) return MCQ_EINVAL;
    if (code_bytes != 1 && code_bytes != 8) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!codes || !prepared || !out) return MCQ_EINVAL;
    mcq_oracle *o = open_state(prepared, 1.0f, N, K, D, 0);
    if (!o) return MCQ_EINVAL;
    /* _maybe_separate_indexes (:551-573): digit t of code j -> codebook j*rep + t */
    mcq_code *idx = (mcq_code *)malloc(sizeof(mcq_code) * (size_t)B * N);
    for (long b = 0; b < B; b++)
        for (int j = 0; j < codes_per_row; j++) {
            size_t at = (size_t)b * codes_per_row + j;
            uint64_t c = code_bytes == 1 ? ((const uint8_t *)codes)[at] : (uint64_t)((const int64_t *)codes)[at];
            for (int t = 0; t < rep; t++) {
                idx[(size_t)b * N + j * rep + t] = (mcq_code)(c % (uint64_t)K);
                c /= (uint64_t)K;
            }
        }
    mcq_oracle_decode(o, idx, B, out);
    free(idx);
    mcq_oracle_free(o);
    return 0;
}

int mcq_logits_host(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                    float *out, void *workspace, size_t workspace_bytes, void *stream) {
    (void)stream; (void)workspace; (void)workspace_bytes;
    if (!domain_ok(N, K, D)) return MCQ_EUNSUPPORTED;
    if (B == 0) return 0;
    if (!x || !prepared || !out || B < 0) return MCQ_EINVAL;
    mcq_oracle *o = open_state(prepared, lscale_exp, N, K, D, 1);
    if (!o) return MCQ_EINVAL;
    int rc = mcq_oracle_logits(o, x, B, out);
    mcq_oracle_free(o);
    return rc;
}
