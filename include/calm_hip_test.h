/*
 * calm_hip_test.h -- unit-level entry points of libcalm_hip_test.so (calm_amd/csrc/test_hooks.hip), used by tests/ and
 * tools/.  The drop-in library libcalm_hip.so does not export them.
 *
 * They run the SAME device kernels as forward_hip (the product source compiled a second time) on caller-provided host
 * buffers (uploaded internally), so a parity failure of a whole decode step can be localised to one kernel.  The
 * reference has no counterpart (it has no tests); semantics cite src/infer.c.
 */
#ifndef CALM_HIP_TEST_H
#define CALM_HIP_TEST_H

#include <stddef.h>
#include <stdint.h>

#include "calm_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* out[d] = W(d,n) . x[n]   -- the row engine without a norm prologue (src/infer.c:209-221).
 * w: d rows of n weights stored as dbits (16 fp16 / 8 fp8-e5m2 / 4 gf4 words). Host pointers. */
void calm_hip_test_matvec(int dbits, const void* w, const float* x, float* out, int n, int d);

/* out[d] = W(d,n) . norm(x)[n], norm = RMSNorm or bias-free LayerNorm with weight nw
 * (src/infer.c:183-207 then :209-221) -- the row engine with its norm prologue. */
void calm_hip_test_norm_matvec(int dbits, const void* w, const float* x, const float* nw, float* out, int n, int d, float eps, int ln);

/* One attention call over a caller-provided fp16 KV cache in the ORACLE's layout
 * [seq_len][kv_dim] (src/infer.c:355-357); it is re-laid out to the backend's private layout
 * internally.  q: (n_heads*head_dim), out: (n_heads*head_dim).  n_split > 1 exercises the
 * split-KV path + merge kernel. */
void calm_hip_test_attn(const float* q, const uint16_t* kcache, const uint16_t* vcache, float* out, int n_heads, int n_kv_heads, int head_dim,
                        int seq_len, int kv_len, int n_split);

/* first index of the strict maximum (reference src/sampler.c:34-42), computed on the device */
int calm_hip_test_argmax(const float* logits, int n);
/* one prompt GEMM alone (prefill.hip.h): out[nb][M] = x[nb][K] . w[M][K]^T, w in a model weight format (dbits 4 / 8 / 16), x as
 * hi + lo binary16.  form 0 / -1 / -3: k_pf_gemm with 2 / 1 / 3 unit strips per wave; 1: k_pf_gemm_wide; 2..8: k_pf_gemm_wide with
 * K cut into that many ranges (one workgroup each, last arriver folds) */
void calm_hip_test_pf_gemm(int dbits, const void* w, const float* x, float* out, int M, int K, int nb, int form);
/* k_sample_minp alone on n host logits: one draw, *rng_state advanced (sampler as in src/sampler.h) */
int calm_hip_test_sample(const float* logits, int n, float temperature, float minp, unsigned long long* rng_state);

/* Re-layout helper: reads the K or V cache of one layer of a transformer prepared by libcalm_hip.so into the oracle's
 * [seq_len][kv_dim] layout (host) as binary16 patterns -- an fp8 cache's e5m2 bytes widened (byte << 8).  which: 0 = K, 1 = V. */
void calm_hip_read_kv(struct Transformer* transformer, int layer, int which, uint16_t* host);

/* the inverse: fills that layer's cache from a [seq_len][kv_dim] array of binary16 patterns (an fp8 cache keeps the top
 * byte), so that a test can start deep inside a long context */
void calm_hip_write_kv(struct Transformer* transformer, int layer, int which, const uint16_t* host);

/* Routing of the LAST decode step at one layer of a mixture-of-experts model prepared by libcalm_hip.so: the n_experts_ac expert
 * ids in rank order and their weights (what src/infer.c:277-305 leaves in moe_experts / moe_weights for that layer).  Read from
 * state.exp (include/calm_hip.h: prepare_hip); a model split over CALM_HIP_DEVICES stages keeps its routing per stage and has none there. */
void calm_hip_read_moe(struct Transformer* transformer, int layer, int* experts, float* weights);

/* Streaming-read micro-benchmark: sums `bytes` of device memory with 16-byte loads
 * (nt != 0: non-temporal) `iters` times; returns GB/s.  bytes <= 128 MiB stays in the 256 MiB
 * Infinity Cache after the first pass, bytes >= 1 GiB measures HBM. */
double calm_hip_membench(size_t bytes, int nt, int iters);

#ifdef __cplusplus
}
#endif

#endif /* CALM_HIP_TEST_H */
