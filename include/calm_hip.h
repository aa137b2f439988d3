/*
 * calm_hip.h -- C ABI of the MI355X (gfx950) infer backend: libcalm_hip.so
 *
 * The first four entry points are exactly what calm's host program binds for a GPU backend
 * (reference src/run.c:22-25 declares upload_cuda / prepare_cuda / forward_cuda / perf_cuda and
 * src/run.c:27-30 the Metal quartet incl. init_metal); they keep the same signatures, argument
 * meaning, call order and error behaviour (no error returns: any HIP failure prints a message
 * and abort()s, like CUDA_CHECK in reference src/infer.cu:12-20).
 *
 * Call order (reference src/run.c:550-612):
 *     [init_hip]  ->  upload_hip(tensor) for every "model.*" tensor  ->  host fills struct Weights
 *     with the returned device pointers  ->  host sets state.kvbits  ->  prepare_hip(t)
 *     ->  t->forward = forward_hip  ->  forward_hip(t, token, pos, flags) once per token.
 *
 * Everything below the "extensions" line is new surface that the reference does not have; the
 * drop-in path never needs it.
 */
#ifndef CALM_HIP_H
#define CALM_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "calm_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the drop-in quartet (+ optional init) ------------------------------------------------ */

/* replaces init_metal (src/run.c:27,566): optional; selects the device (env CALM_HIP_DEVICE or
 * LOCAL_RANK, default 0) and creates the stream. upload_hip/prepare_hip call it lazily. */
void init_hip(void);

/* replaces upload_cuda (src/run.c:22,557; src/infer.cu:69-71): copies `size` bytes from `host`
 * into freshly allocated device memory and returns the DEVICE pointer, which the host stores
 * back into tensor->data. The copy is complete on return (it goes through two pinned staging buffers on the
 * backend's stream, the host's copy into one overlapping the DMA out of the other: 7 GB in 0.3 s).  With CALM_HIP_DEVICES > 1
 * the copy is deferred to prepare_hip and the mmap must stay mapped until then (it does: src/run.c:515,637). */
void* upload_hip(void* host, size_t size);

/* extension (no reference counterpart): `size` bytes of device memory with the slack behind it that this backend's kernels
 * rely on (their 16-byte loads are unclamped at the end of a row or vector) -- for hosts that PRODUCE a tensor on the device
 * instead of uploading it (tools/synth_fill_hip.hip: synthetic fixtures, an on-device gf4 quantiser).  A pointer from
 * upload_hip or alloc_hip is what struct Weights may point at; plain hipMalloc'ed memory is not.  Freed with free_hip. */
void* alloc_hip(size_t size);

/* replaces prepare_cuda (src/run.c:23,580; src/infer.cu:73-131): allocates activations, the KV
 * cache (state.kvbits must already be 8 or 16; for head size 128, windows of whole 64-position blocks longer than "split_min" positions and the knob
 * "attn_vt" on at this point, the VALUE cache is kept twice, the second copy transposed for the matrix-core attention of long contexts -- + 50 % of
 * the KV cache's memory; state.value_cache then points at an allocation twice the reference's size; if that does not fit, the backend runs without it) and the host-visible logits buffer, and snapshots
 * the per-layer weight pointers. Fills state.x/hb/he/q/att/key_cache/value_cache/logits, and -- an extension, the reference's
 * GPU backend leaves it unset -- state.exp: device memory holding the routing of the last decode step, [n_layers][CALM_MAX_EXPERTS]
 * float weights in rank order followed by as many int expert ids (dense models: weight 1, expert 0).
 * Mixture-of-experts models also get, per layer, a [dim][n_experts] fp32 table derived from moegate and the FFN norm weight (the router's
 * logits are accumulated by the attention output projection's epilogue: 4 MB for Mixtral-8x7B, 16 MB for DBRX).
 * Aborts (like every error here) on shapes outside the backend's limits: dbits 4/8/16; dim, hidden_dim and
 * n_heads*head_dim multiples of 128/dbits; head_dim a multiple of 8, at most 512; and -- the one limit the reference's
 * backends do not have -- dim and n_heads*head_dim must fit one CU's 160 KiB LDS as an fp32 vector (about 40K elements at
 * fp16 / fp8, 36K at gf4): the matvec kernels stage their whole input vector there.  hidden_dim may be wider: the FFN
 * down-projection then runs as several launches over column ranges. */
void prepare_hip(struct Transformer* transformer);

/* replaces forward_cuda (src/run.c:24,581; src/infer.cu:743-759): one decode step for `token`
 * at absolute position `pos` (may exceed seq_len: rolling KV buffer with CALM_KV_SINKS sinks;
 * may go backwards between calls). Returns state.logits (vocab_size floats, host memory, valid
 * until the next call, caller may overwrite) after synchronising; with FF_UPDATE_KV_ONLY returns
 * NULL after enqueueing the work without synchronising (src/infer.cu:724-727). */
float* forward_hip(struct Transformer* transformer, int token, int pos, unsigned flags);

/* replaces perf_cuda (src/run.c:25,629-633; src/infer.cu:761-801): prints a per-stage
 * time / algorithmic-GB/s table accumulated while env CALM_HIP_PROF=1; otherwise a no-op. */
void perf_hip(void);

/* ---- extensions (not needed by the drop-in path) ------------------------------------------ */

/* number of visible HIP devices (0 => no GPU; every other entry point aborts in that case) */
int calm_hip_device_count(void);

/* frees everything prepare_hip allocated for this transformer (the reference never frees,
 * src/run.c:636; tests load many models per process). Uploaded weights are NOT freed here. */
void release_hip(struct Transformer* transformer);

/* frees one buffer returned by upload_hip */
void free_hip(void* device);

/* extension: copy `size` bytes of device memory (e.g. state.x, an uploaded tensor) back to the host, after draining the
 * backend's stream */
void download_hip(void* host, const void* device, size_t size);

/* Greedy decode of n_steps tokens entirely on the device (argmax on the GPU, next token fed
 * back without a host round trip); equals n_steps calls of forward_hip + sample_argmax
 * (reference src/sampler.c:34-42: first index of the strict maximum). Writes the n_steps sampled
 * token ids to out_tokens; returns state.logits of the last step. */
float* decode_greedy_hip(struct Transformer* transformer, int token, int pos, int n_steps, int* out_tokens);

/* The same with the reference's sampler on the device: n_steps tokens, each drawn as sample(sampler, logits) would draw it
 * (src/sampler.c:80-90: greedy when temperature == 0 or minp >= 1, else one xorshift* coin and the min-p cut of
 * src/sampler.c:44-78), the draw fed back as the next token without a host round trip.  sampler->rng_state comes back
 * advanced by the coins that were drawn, as the host loop would leave it.  Against the host sampler on the same logits the
 * draw can differ only when the coin lands within rounding of a boundary between two surviving tokens (the device's expf
 * and a bracketed running sum; the reference's own sum is compiled -ffast-math). */
float* decode_sample_hip(struct Transformer* transformer, int token, int pos, int n_steps, int* out_tokens, struct Sampler* sampler);

/* Batched prompt ingestion: the KV-cache effect of
 *     for (i = 0; i < n; ++i) forward_hip(transformer, tokens[i], pos + i, FF_UPDATE_KV_ONLY);
 * i.e. of the reference's serial prompt loop (src/run.c:208,216-218; README.md:80 "prompt processing is
 * serial"), computed up to 2048 tokens at a time (mixture-of-experts models: 4096): weights are streamed once per chunk and the multiply-adds run on
 * the f16 matrix cores with the fp32 activations carried as hi + lo binary16 (every product exact, fp32 accumulation:
 * 3-7e-7 per GEMM), so the cache rows agree with the serial path to fp32 rounding.  A chunk in which an activation leaves the binary16 range (beyond +-65504, or NaN)
 * is redone token by token through the serial fp32 decode path inside the call (calm_hip_query("pf_redone") counts such tokens).
 * Returns after the work is complete (`tokens` is host memory and may be reused).
 * Mixture-of-experts models are routed per token on the device and each expert runs one GEMM over the rows
 * routed to it.  Positions at or beyond seq_len (rolling buffer, sink rotation between tokens) are processed
 * through the decode path one token at a time inside the call -- same result, no speed-up.  On a model sharded inside the
 * library (CALM_HIP_DEVICES) a chunk runs stage after stage, its residual rows crossing as one token's x does. */
void prefill_hip(struct Transformer* transformer, const int* tokens, int n, int pos);

/* The same, plus the model's verdict on the text: logprob[i] = log softmax(logits after tokens[i])[tokens[i + 1]]
 * for i < n - 1 (logprob[n - 1] = 0) -- the quantity the reference's perplexity mode accumulates one forward()
 * at a time (src/run.c:294-298, sample_prob src/sampler.c:19-32); final norm + classifier run as one more GEMM
 * per 256-token chunk.  `logprob` is host memory, n floats. */
void prefill_logprobs_hip(struct Transformer* transformer, const int* tokens, int n, int pos, float* logprob);

/* Layer-pipeline stage (for models beyond one GPU's 288 GB): `transformer` describes
 * only THIS stage's slice of the model -- config.n_layers = the stage's layer count, weights indexed from 0,
 * token_embedding_table set on the first stage only, rms_final_weight / wcls on the last stage only.
 *   CALM_STAGE_FIRST : embed `token` into state.x; otherwise state.x must already hold the residual stream
 *                      received from the previous stage (copy it in with copy_hip).
 *   CALM_STAGE_LAST  : final norm + classifier; returns state.logits.  Other stages return NULL.
 * Always synchronises before returning, so state.x (dim floats, device memory) can be sent on.
 * forward_hip(t, ...) == forward_stage_hip(t, ..., CALM_STAGE_FIRST | CALM_STAGE_LAST). */
enum CalmHipStageFlags {
	CALM_STAGE_FIRST = 1 << 0,
	CALM_STAGE_LAST = 1 << 1,
};
float* forward_stage_hip(struct Transformer* transformer, int token, int pos, unsigned flags, unsigned stage_flags);

/* synchronous copy between any two of {device, host} buffers (hipMemcpyDefault); used to move state.x in and
 * out of the communication buffers of a pipeline */
void copy_hip(void* dst, const void* src, size_t size);

/* Stage timer for roofline reporting: launches the kernel of `stage` for every layer in turn
 * (so successive launches stream different weights and cannot hit the 256 MiB Infinity Cache),
 * `iters` sweeps, bracketed by hipEvents on the backend's own stream.
 * Returns the average duration of ONE launch in microseconds and stores the algorithmic bytes
 * one launch reads (reference accounting, src/infer.cu:692-699) in *bytes_per_launch. */
enum CalmHipStage {
	CALM_STAGE_QKV = 0,      /* norm + wq/wk/wv matvec + bias + clip + RoPE + KV append */
	CALM_STAGE_ATTN = 1,     /* KV-cache attention (online softmax) */
	CALM_STAGE_ATTN_OUT = 2, /* wo matvec + residual */
	CALM_STAGE_FFN_UP = 3,   /* norm + [moe gate] + act(w1 x) * (w3 x) */
	CALM_STAGE_FFN_DOWN = 4, /* w2 matvec + weighted residual */
	CALM_STAGE_OUTPUT = 5,   /* final norm + classifier */
	CALM_STAGE_COUNT = 6,
};
double perf_stage_hip(struct Transformer* transformer, int stage, int iters, uint64_t* bytes_per_launch);

/* name of the HIP device in use (static storage) */
const char* calm_hip_device_name(void);

/* Run-time knobs -- twelve keys, "stage" included (the CALM_HIP_GRAPH / _PROF / _SPLIT_T / _SPLIT_MIN / _ATTN_VT / _QKV_ATTN / _PF_CHUNK /
 * _PF_CHUNK_MOE / _PF_SCORE_MB environment variables read by init_hip set the same switches before the first model):
 *   "graph"     1 = replay each step from a hipGraph (default), 0 = eager launches
 *   "prof"      1 = eager launches bracketed by per-stage events, reported by perf_hip (a model split over stages: also an event pair
 *               around every stage-to-stage copy)
 *   "split_t"   cached positions per attention KV split (default 128)
 *   "split_min" contexts up to this many positions are not split (default 384)
 *   "attn_vt"   1 = split attention on the matrix cores over the transposed value cache (default; head size 128), 0 = lane arithmetic.
 *               Read by prepare_hip: 0 at that point also saves the transposed cache's memory (+ 50 % of the KV cache)
 *   "qkv_attn"  1 = a step whose cached rows fit one workgroup's registers (256 at head size 128, 512 at 64; fp8 / fp16 weights) runs
 *               its attention inside the QKV projection's launch (default), 0 = two launches
 *   "forms", "pf_forms": bit sets that force kernel forms the launchers otherwise pick by shape (0 = the rules; the forms are listed at
 *               their definition in calm_amd/csrc/infer_hip.hip) -- for the tests that run every shipping form on small fixtures
 *   "pf_chunk": tokens per prompt chunk of a dense model, 1024 ... 2048 in steps of 128 (default 2048; read when a model's prompt
 *       buffers are allocated, i.e. at its first prefill_hip call; never more than the context window rounded up to 128)
 *   "pf_chunk_moe": ... of a mixture-of-experts model: 1024, 2048 or 4096 (default 4096; read at the same moment).  The experts'
 *       gathered rows take chunk x active experts x (dim + hidden_dim) x 4 bytes of scratch: 0.8 GB for Mixtral-8x7B, 1.7 GB for
 *       DBRX-132B at 4096, a quarter of that at 1024
 *   "pf_score_mb": MiB of device scratch for the logits of prefill_logprobs_hip (default 256; read when that scratch is allocated, at a
 *       model's first scoring call): a chunk is scored in blocks of as many tokens as fit (whole 128-token columns, at least 128)
 *   "stage"     multi-device placement: route the following upload_hip / alloc_hip calls to that stage's device (-1: defer to
 *               prepare_hip; -1 is a value here, not a query)
 * value < 0 only queries.  Returns the previous value, or -1 for an unknown key.  Changing a knob that shapes the launches drops
 * the captured graphs of every prepared model (they are re-captured on next use). */
int calm_hip_configure(const char* key, int value);

/* Read-only state (nothing here changes a launch): "stages" (pipeline stages of this process), "stage_device" (value = stage -> its
 * device), "pf_redone" (prompt tokens prefill_hip sent back through the serial path), "handoffs" / "handoff_ns" (stage-to-stage copies
 * timed under "prof" and their average duration), "fused_steps" (decode steps that took the "qkv_attn" launch), "fuse_timeouts" (its
 * bounded waits that expired: 0 unless a launch lost a producer; the affected head's output is NaN).  -1 for an unknown key. */
int calm_hip_query(const char* key, int value);

#ifdef __cplusplus
}
#endif

#endif /* CALM_HIP_H */
