/*
 * calm_abi.h -- the model ABI shared between calm's host program and an infer backend.
 *
 * This header restates, field for field, the binary layout of the reference's
 * `struct Config / Weights / RunState / Transformer` (reference src/model.h:12-85) so that a
 * `struct Transformer*` produced by the reference's run.c (src/run.c:520-596) can be handed to
 * this backend unchanged, and so that our own hosts (C++ CLI, Python/ctypes) can build one.
 * The layout is the contract; tests/test_abi.py checks every offset against the reference
 * header when /root/reference is present, and against frozen numbers otherwise.
 *
 * Plain C, no HIP or torch types: this is the drop-in boundary.
 */
#ifndef CALM_ABI_H
#define CALM_ABI_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/model.h:6-10 */
#define CALM_MAX_LAYERS 128
#define CALM_MAX_EXPERTS 64
#define CALM_KV_SINKS 2 /* attention sinks kept at the head of the rolling KV buffer */

/* hyper-parameters; filled from .calm metadata by the host (reference src/run.c:32-69) */
struct Config {
	int dim;
	int hidden_dim;
	int head_dim;
	int n_layers;
	int n_heads;
	int n_kv_heads;
	int vocab_size;
	int seq_len;
	float rope_theta;
	int rotary_dim;
	int n_experts;
	int n_experts_ac;
	float norm_eps;
	bool act_gelu; /* GELU-tanh instead of SiLU */
	bool norm_ln;  /* mean-subtracting LayerNorm (no bias) instead of RMSNorm */
	bool norm_par; /* parallel residual: FFN reuses the attention-norm output */
	float qkv_clip;
};

/* weight pointers; after upload_hip these are DEVICE pointers (reference src/run.c:550-576).
 * dbits selects the storage type behind every void*: 16 = fp16, 8 = fp8 e5m2, 4 = gf4 words. */
struct Weights {
	int dbits;
	void* token_embedding_table;            /* (vocab, dim) */
	float* rms_att_weight[CALM_MAX_LAYERS]; /* (dim) */
	float* rms_ffn_weight[CALM_MAX_LAYERS]; /* (dim); NULL for norm_par models */
	void* wq[CALM_MAX_LAYERS];              /* (n_heads*head_dim, dim) */
	void* wk[CALM_MAX_LAYERS];              /* (n_kv_heads*head_dim, dim) */
	void* wv[CALM_MAX_LAYERS];              /* (n_kv_heads*head_dim, dim) */
	void* wo[CALM_MAX_LAYERS];              /* (dim, n_heads*head_dim) */
	void* w1[CALM_MAX_LAYERS];              /* ([n_experts,] hidden, dim) */
	void* w2[CALM_MAX_LAYERS];              /* ([n_experts,] dim, hidden) */
	void* w3[CALM_MAX_LAYERS];              /* ([n_experts,] hidden, dim) */
	float* rms_final_weight;                /* (dim) */
	void* wcls;                             /* (vocab, dim); == token_embedding_table when tied */
	float* bqkv[CALM_MAX_LAYERS];           /* ((n_heads + 2*n_kv_heads)*head_dim) or NULL */
	void* moegate[CALM_MAX_LAYERS];         /* (n_experts, dim) or NULL */
};

/* activation buffers + KV cache; owned by the backend after prepare_hip */
struct RunState {
	float* x;
	float* xb;
	float* xb2;
	float* hb;
	float* hb2;
	float* he;
	float* q;
	float* k;
	float* v;
	float* att;
	float* exp;
	float* logits; /* host-readable AND host-writable (the sampler mutates it) */
	int kvbits;    /* set by the host BEFORE prepare: 16 = fp16 cache, 8 = fp8 e5m2 cache */
	void* key_cache;
	void* value_cache;
};

struct Transformer {
	struct Config config;
	struct Weights weights;
	struct RunState state;
	size_t n_params, n_bytes, n_bandwidth;
	float* (*forward)(struct Transformer* transformer, int token, int pos, unsigned flags);
};

/* reference src/model.h:87-89 */
enum ForwardFlags {
	FF_UPDATE_KV_ONLY = 1 << 0, /* run every layer (KV append included) but skip final norm + classifier; return NULL */
};

/* reference src/sampler.h:3-9 -- what the host program hands to sample(); decode_sample_hip takes the same object */
struct Sampler {
	int vocab_size;
	unsigned long long rng_state; /* xorshift* state, advanced once per sampled token (src/sampler.c:7-18,84) */
	float temperature;            /* 0 => greedy */
	float minp;                   /* >= 1 => greedy */
};

#ifdef __cplusplus
}
#endif

#endif /* CALM_ABI_H */
