/*
 * mcq_host.h -- host twins of include/mcq.h: the same entry points with the same signatures, error
 * codes and argument checks, taking HOST pointers and running the CPU oracle (mcq_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY (exported by oracle/_build/libmcq_oracle.so, never by libmcq_hip.so and
 * never loaded by quantization_amd/): the parity tests push one argument list through
 * mcq_encode/mcq_decode on device pointers and through mcq_encode_host/mcq_decode_host on host
 * copies and compare the outputs bit for bit.  `stream` and `workspace` are accepted and ignored.
 */
#ifndef MCQ_HOST_H
#define MCQ_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

size_t mcq_prepared_bytes_host(int N, int K, int D);
int mcq_prepare_host(const float *centers, float cscale_exp, const float *weight, const float *bias,
                     int N, int K, int D, void *prepared, void *stream);
size_t mcq_encode_workspace_bytes_host(long B, int N, int K, int D);
int mcq_encode_host(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                    int refine_iters, uint8_t *out_u8, int64_t *out_i64, void *workspace,
                    size_t workspace_bytes, void *stream);
int mcq_refine_indexes_host(const float *x, long B, const void *prepared, int N, int K, int D, int refine_iters,
                            const int64_t *idx_in, int64_t *idx_out, void *workspace, size_t workspace_bytes,
                            void *stream);
int mcq_decode_host(const void *codes, int code_bytes, int codes_per_row, long B, const void *prepared,
                    int N, int K, int D, float *out, void *stream);
int mcq_logits_host(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                    float *out, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
