/*
 * calm_oracle.c -- CPU restatement of calm's single-batch decode step.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle for the HIP backend.  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * path (libcalm_hip.so) never links, calls or falls back to anything in oracle/.
 *
 * It restates, in plain portable C (no intrinsics, no _Float16, builds with gcc 11), the
 * algorithm of the reference CPU backend src/infer.c; every function cites the lines it
 * follows.  Summation is the reference's scalar path (the `#else` branches of
 * src/infer.c:62-69,92-99,129-139): fp32 accumulate, sequential over the row.
 *
 * Pinning: the reference ships no golden vectors or tests for this path,
 * so the oracle is pinned against the reference ITSELF: oracle/Makefile builds the untouched
 * src/infer.c into oracle/_ref/libcalm_ref.so and tests/golden/ holds logits that library
 * produced for small synthetic .calm models (generator: tests/golden/make_golden.py);
 * tests/test_oracle.py checks this file against both.
 *
 * The ABI structs are shared with the backend (include/calm_abi.h == reference src/model.h).
 */
#include <assert.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/calm_abi.h"

/* ---------------------------------------------------------------- storage formats ---------- */

/* IEEE binary16 -> binary32, exact.  The reference relies on the compiler's _Float16
 * (src/infer.c:18-23); this is the same mapping written out. */
float oracle_half_to_float(uint16_t h) {
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1f;
	uint32_t man = h & 0x3ffu;
	uint32_t bits;
	if (exp == 0) {
		if (man == 0) {
			bits = sign;
		} else {
			/* subnormal: renormalise */
			int e = -1;
			do {
				man <<= 1;
				e++;
			} while ((man & 0x400u) == 0);
			bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
		}
	} else if (exp == 31) {
		bits = sign | 0x7f800000u | man << 13;
	} else {
		bits = sign | (exp + 127 - 15) << 23 | man << 13;
	}
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

/* binary32 -> binary16, round-to-nearest-even (what `(half)x` / F16C vcvtps2ph do for the KV
 * cache store, src/infer.c:378-381). */
uint16_t oracle_float_to_half(float f) {
	uint32_t x;
	memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t absx = x & 0x7fffffffu;
	if (absx >= 0x7f800000u) { /* inf / nan */
		return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? (0x200u | ((absx >> 13) & 0x3ffu)) : 0));
	}
	if (absx >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
		return (uint16_t)(sign | 0x7c00u);
	}
	if (absx < 0x33000001u) { /* <= 2^-25 -> +-0 (exactly 2^-25 ties to even = 0) */
		return (uint16_t)sign;
	}
	int e = (int)(absx >> 23) - 127;
	uint32_t man = (absx & 0x7fffffu) | 0x800000u;
	int shift; /* bits to drop from the 24-bit significand */
	uint32_t base;
	if (e < -14) { /* subnormal half */
		shift = 13 + (-14 - e);
		base = 0;
	} else {
		shift = 13;
		base = (uint32_t)(e + 15) << 10;
		man &= 0x7fffffu;
	}
	uint32_t q = man >> shift;
	uint32_t rem = man & ((1u << shift) - 1);
	uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (q & 1))) {
		q++;
	}
	return (uint16_t)(sign | (base + q)); /* a carry out of the mantissa correctly bumps the exponent */
}

/* the same mapping as a table, filled from the function above when the library is loaded: the hot loops below convert a
 * binary16 per multiply-add, and 4096-8192-position attention tests spent minutes in the branches of the written-out form */
static float h2f_table[65536];
__attribute__((constructor)) static void h2f_init(void) {
	for (uint32_t i = 0; i < 65536; ++i) {
		h2f_table[i] = oracle_half_to_float((uint16_t)i);
	}
}
static inline float h2f(uint16_t h) {
	return h2f_table[h];
}

/* fp8 e5m2 is the top byte of the binary16 pattern (src/infer.c:28-35) */
static inline float fp8_to_float(uint8_t v) {
	return h2f((uint16_t)((uint16_t)v << 8));
}

/* binary32 -> fp8 e5m2, ONE round-to-nearest-even step from the fp32 value, saturating to the largest finite code
 * (0x7b = 57344) on overflow and infinity, NaN kept: `__nv_fp8_e5m2(float)` of the reference's CUDA backend
 * (__NV_SATFINITE), which is how it stores K/V rows when kvbits == 8 (src/infer.cu:473-482, sink keys :150-180).
 * The reference's CPU backend has no fp8 cache (src/infer.c:161 asserts kvbits == 16): this mode restates the CUDA
 * path's STORAGE on top of the CPU path's arithmetic so that the HIP backend's fp8 cache has something to answer to. */
uint8_t oracle_float_to_e5m2(float f) {
	uint32_t x;
	memcpy(&x, &f, 4);
	uint32_t sign = (x >> 24) & 0x80u;
	uint32_t absx = x & 0x7fffffffu;
	if (absx > 0x7f800000u) {
		return (uint8_t)(sign | 0x7fu);
	}
	if (absx >= 0x47600000u) { /* >= 57344: every such value rounds to or beyond the largest finite code */
		return (uint8_t)(sign | 0x7bu);
	}
	int e = (int)(absx >> 23) - 127;
	if (e < -18) { /* below half of the smallest subnormal (2^-16): zero */
		return (uint8_t)sign;
	}
	uint32_t man = (absx & 0x7fffffu) | 0x800000u;
	int shift;
	uint32_t base;
	if (e < -14) { /* subnormal e5m2 */
		shift = 21 + (-14 - e);
		base = 0;
	} else {
		shift = 21;
		base = (uint32_t)(e + 15) << 2;
		man &= 0x7fffffu;
	}
	uint32_t q = man >> shift;
	uint32_t rem = man & ((1u << shift) - 1);
	uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (q & 1))) {
		q++;
	}
	uint32_t r = base + q; /* a carry out of the mantissa bumps the exponent */
	return (uint8_t)(sign | (r > 0x7bu ? 0x7bu : r));
}

/* what a K/V element becomes in the cache: binary16 RNE (src/infer.c:378-381), or -- kvbits == 8 -- e5m2, kept in the same
 * 16-bit slot as the binary16 pattern of the e5m2 value (byte << 8), so every reader below is unchanged */
static inline uint16_t kv_store(float f, int kvbits) {
	return kvbits == 8 ? (uint16_t)((uint16_t)oracle_float_to_e5m2(f) << 8) : oracle_float_to_half(f);
}

/* gf4: 32-bit word = fp8 scale in bits 0-7 and eight 3-bit codes; w_k = (code_k - 4) * scale / -4
 * (src/infer.c:37-40) */
static inline float gf4_to_float(uint32_t word, int k) {
	float s = fp8_to_float((uint8_t)(word & 0xff)) / -4.f;
	return (float)((int)((word >> (8 + k * 3)) & 7) - 4) * s;
}

float oracle_decode_weight(const void* w, int dbits, size_t idx) {
	switch (dbits) {
	case 16:
		return h2f(((const uint16_t*)w)[idx]);
	case 8:
		return fp8_to_float(((const uint8_t*)w)[idx]);
	case 4:
		return gf4_to_float(((const uint32_t*)w)[idx / 8], (int)(idx % 8));
	default:
		assert(!"dbits must be 4, 8 or 16");
		return 0.f;
	}
}

/* ---------------------------------------------------------------- row dot products --------- */

/* src/infer.c:44-70 (scalar branch) */
static float dot_fp16(const void* w, int n, int i, const float* x) {
	const uint16_t* r = (const uint16_t*)w + (size_t)i * n;
	float val = 0.0f;
	for (int j = 0; j < n; j++) {
		val += h2f(r[j]) * x[j];
	}
	return val;
}

/* src/infer.c:72-100 (scalar branch) */
static float dot_fp8(const void* w, int n, int i, const float* x) {
	const uint8_t* r = (const uint8_t*)w + (size_t)i * n;
	float val = 0.0f;
	for (int j = 0; j < n; j++) {
		val += fp8_to_float(r[j]) * x[j];
	}
	return val;
}

/* src/infer.c:102-140 (scalar branch) */
static float dot_gf4(const void* w, int n, int i, const float* x) {
	const uint32_t* r = (const uint32_t*)w + (size_t)i * n / 8;
	float val = 0.0f;
	for (int j = 0; j < n; j += 8) {
		uint32_t wg = r[j / 8];
		for (int k = 0; k < 8; ++k) {
			val += gf4_to_float(wg, k) * x[j + k];
		}
	}
	return val;
}

typedef float (*dot_fn)(const void* w, int n, int i, const float* x);

static dot_fn pick_dot(int dbits) {
	assert(dbits == 4 || dbits == 8 || dbits == 16);
	return dbits == 4 ? dot_gf4 : (dbits == 8 ? dot_fp8 : dot_fp16);
}

/* W (d,n) @ x (n) [+ b] -> out (d); src/infer.c:209-221 */
void oracle_matvec(float* out, const float* x, const void* w, const float* b, int n, int d, int dbits) {
	dot_fn dot = pick_dot(dbits);
	int i;
#pragma omp parallel for private(i)
	for (i = 0; i < d; i++) {
		float val = dot(w, n, i, x);
		if (b) {
			val += b[i];
		}
		out[i] = val;
	}
}

/* ---------------------------------------------------------------- small ops ---------------- */

/* RMSNorm, or bias-free LayerNorm when ln; src/infer.c:183-207.  o may alias x. */
void oracle_norm(float* o, const float* x, const float* weight, int size, float eps, int ln) {
	float mean = 0.0f;
	if (ln) {
		for (int j = 0; j < size; j++) {
			mean += x[j];
		}
		mean /= size;
	}
	float ss = 0.0f;
	for (int j = 0; j < size; j++) {
		ss += (x[j] - mean) * (x[j] - mean);
	}
	float var = ss / size;
	float scale = 1.0f / sqrtf(var + eps);
	for (int j = 0; j < size; j++) {
		o[j] = (x[j] - mean) * scale * weight[j];
	}
}

/* interleaved-pair RoPE over d elements made of heads of head_dim; src/infer.c:223-236 */
void oracle_rope(float* vec, int d, int head_dim, int pos, float theta, int rotary_dim) {
	for (int i = 0; i < d; i += 2) {
		int j_head = i % head_dim;
		float freq = j_head >= rotary_dim ? 0.f : 1.0f / powf(theta, (float)j_head / (float)rotary_dim);
		float val = pos * freq;
		float fcr = cosf(val);
		float fci = sinf(val);
		float v0 = vec[i];
		float v1 = vec[i + 1];
		vec[i] = v0 * fcr - v1 * fci;
		vec[i + 1] = v0 * fci + v1 * fcr;
	}
}

/* one head of attention over kv_len cached positions (fp16 K/V rows kv_dim apart);
 * src/infer.c:238-267 */
void oracle_attn_head(float* xout, float* atth, const float* qh, const uint16_t* kh, const uint16_t* vh, int head_dim, int kv_dim, int kv_len) {
	float score_max = -FLT_MAX;
	for (int t = 0; t < kv_len; ++t) {
		float score = 0.0f;
		for (int j = 0; j < head_dim; ++j) {
			score += qh[j] * h2f(kh[(size_t)t * kv_dim + j]);
		}
		score /= sqrtf((float)head_dim);
		score_max = (score_max < score) ? score : score_max;
		atth[t] = score;
	}
	float score_sum = 0.f;
	for (int t = 0; t < kv_len; ++t) {
		atth[t] = expf(atth[t] - score_max);
		score_sum += atth[t];
	}
	for (int j = 0; j < head_dim; ++j) {
		float res = 0.f;
		for (int t = 0; t < kv_len; ++t) {
			res += (atth[t] / score_sum) * h2f(vh[(size_t)t * kv_dim + j]);
		}
		xout[j] = res;
	}
}

/* src/infer.c:269-275 */
static inline float act_gelu(float x) {
	return 0.5f * x * (1.0f + tanhf(0.797885f * (x + 0.044715f * x * x * x)));
}
static inline float act_silu(float x) {
	return x / (1.0f + expf(-x));
}
float oracle_act(float x, int gelu) {
	return gelu ? act_gelu(x) : act_silu(x);
}

/* top-`active` experts by gate logit, ties to the lowest index, weights = softmax over the
 * selected logits only; src/infer.c:277-305 */
void oracle_moe_gate(float* moe_weights, int* moe_experts, const float* x, int d, int active) {
	float max_val = -FLT_MAX;
	for (int j = 0; j < d; ++j) {
		max_val = (max_val < x[j]) ? x[j] : max_val;
	}
	uint64_t mask = 0;
	float wsum = 0.0f;
	for (int k = 0; k < active; ++k) {
		int best = -1;
		for (int j = 0; j < d; ++j) {
			if ((mask & (1ull << j)) == 0 && (best == -1 || x[j] > x[best])) {
				best = j;
			}
		}
		moe_experts[k] = best;
		wsum += expf(x[best] - max_val);
		mask |= 1ull << best;
	}
	for (int k = 0; k < active; ++k) {
		moe_weights[k] = expf(x[moe_experts[k]] - max_val) / wsum;
	}
}

static inline float clipf(float x, float v) {
	return x < -v ? -v : (x > v ? v : x);
}

/* rolling KV buffer with attention sinks; src/infer.c:329-332 */
void oracle_kv_slots(int pos, int seq_len, int* kv_sink, int* kv_pos, int* kv_len) {
	*kv_sink = pos >= seq_len ? CALM_KV_SINKS : 0;
	*kv_pos = *kv_sink + (pos - *kv_sink) % (seq_len - *kv_sink);
	*kv_len = pos >= seq_len ? seq_len : pos + 1;
}

/* ---------------------------------------------------------------- whole step --------------- */

/* Test hook: where oracle_forward leaves EVERY layer's routing -- expert ids in rank order [layer][n_experts_ac], their weights,
 * and the gate logits [layer][n_experts] -- for the full-depth routed-expert identity tests.  The reference keeps only the last
 * layer's in state.exp (src/infer.c:423-432); NULL pointers switch the trace off. */
static int* trace_experts;
static float* trace_weights;
static float* trace_logits;
void oracle_trace_moe(int* experts, float* weights, float* logits) {
	trace_experts = experts, trace_weights = weights, trace_logits = logits;
}

/* allocate activations + fp16 KV cache on the host; src/infer.c:142-181 */
void oracle_prepare(struct Transformer* t) {
	struct Config* p = &t->config;
	struct RunState* s = &t->state;
	int q_dim = p->head_dim * p->n_heads;
	int kv_dim = p->head_dim * p->n_kv_heads;
	int nact = p->n_experts_ac ? p->n_experts_ac : 1;

	assert(s->kvbits == 16 || s->kvbits == 8); /* 16: the CPU path's cache (src/infer.c:26,161); 8: the CUDA path's e5m2 rows, see kv_store */
	s->x = calloc(p->dim, sizeof(float));
	s->xb = calloc(p->dim, sizeof(float));
	s->xb2 = calloc(q_dim > p->dim ? q_dim : p->dim, sizeof(float));
	s->hb = calloc(p->hidden_dim > p->dim ? p->hidden_dim : p->dim, sizeof(float));
	s->hb2 = calloc(p->hidden_dim, sizeof(float));
	s->q = calloc(q_dim, sizeof(float));
	s->k = calloc(kv_dim, sizeof(float));
	s->v = calloc(kv_dim, sizeof(float));
	s->att = calloc((size_t)p->n_heads * p->seq_len, sizeof(float));
	s->exp = calloc(p->n_experts + nact * 2, sizeof(float));
	s->logits = calloc(p->vocab_size, sizeof(float));
	s->key_cache = calloc((size_t)p->n_layers * p->seq_len * kv_dim, sizeof(uint16_t));
	s->value_cache = calloc((size_t)p->n_layers * p->seq_len * kv_dim, sizeof(uint16_t));
	if (!s->x || !s->xb || !s->xb2 || !s->hb || !s->hb2 || !s->q || !s->k || !s->v || !s->att || !s->exp || !s->logits || !s->key_cache || !s->value_cache) {
		fprintf(stderr, "calm_oracle: allocation failed\n");
		abort();
	}
}

void oracle_release(struct Transformer* t) {
	struct RunState* s = &t->state;
	free(s->x), free(s->xb), free(s->xb2), free(s->hb), free(s->hb2), free(s->q), free(s->k), free(s->v);
	free(s->att), free(s->exp), free(s->logits), free(s->key_cache), free(s->value_cache);
	memset(s, 0, sizeof(*s));
}

/* One decode step; src/infer.c:311-472.  Weights are HOST pointers here.
 * stage_flags (bit 0 = first stage: embed the token, bit 1 = last stage: final norm + classifier) let the
 * same walk serve as one stage of a layer pipeline whose `t` describes only that stage's layers; the
 * reference has no such split -- with both bits set this IS its forward(). */
float* oracle_forward_stage(struct Transformer* t, int token, int pos, unsigned flags, unsigned stage_flags) {
	struct Config* p = &t->config;
	struct Weights* w = &t->weights;
	struct RunState* s = &t->state;
	float* x = s->x;
	int dim = p->dim;
	int hidden_dim = p->hidden_dim;
	int q_dim = p->head_dim * p->n_heads;
	int kv_dim = p->head_dim * p->n_kv_heads;
	int kv_mul = p->n_heads / p->n_kv_heads;
	int dbits = w->dbits;
	int nact = p->n_experts_ac ? p->n_experts_ac : 1;

	int kv_sink, kv_pos, kv_len;
	oracle_kv_slots(pos, p->seq_len, &kv_sink, &kv_pos, &kv_len);

	/* embedding row -> x; src/infer.c:334-347 */
	if (stage_flags & 1u) {
		for (int i = 0; i < dim; ++i) {
			x[i] = oracle_decode_weight(w->token_embedding_table, dbits, (size_t)token * dim + i);
		}
	}

	for (int l = 0; l < p->n_layers; l++) {
		oracle_norm(s->xb, x, w->rms_att_weight[l], dim, p->norm_eps, p->norm_ln); /* :352 */

		size_t loff = (size_t)l * p->seq_len * kv_dim;
		uint16_t* kb = (uint16_t*)s->key_cache + loff;
		uint16_t* vb = (uint16_t*)s->value_cache + loff;

		/* :360-362 */
		oracle_matvec(s->q, s->xb, w->wq[l], w->bqkv[l], dim, q_dim, dbits);
		oracle_matvec(s->k, s->xb, w->wk[l], w->bqkv[l] ? w->bqkv[l] + q_dim : NULL, dim, kv_dim, dbits);
		oracle_matvec(s->v, s->xb, w->wv[l], w->bqkv[l] ? w->bqkv[l] + q_dim + kv_dim : NULL, dim, kv_dim, dbits);

		/* :365-371 */
		for (int i = 0; i < q_dim; i++) {
			s->q[i] = clipf(s->q[i], p->qkv_clip);
		}
		for (int i = 0; i < kv_dim; i++) {
			s->k[i] = clipf(s->k[i], p->qkv_clip);
			s->v[i] = clipf(s->v[i], p->qkv_clip);
		}

		/* :374-375 */
		oracle_rope(s->q, q_dim, p->head_dim, pos, p->rope_theta, p->rotary_dim);
		oracle_rope(s->k, kv_dim, p->head_dim, pos, p->rope_theta, p->rotary_dim);

		/* :378-381 */
		for (int i = 0; i < kv_dim; i++) {
			kb[(size_t)kv_pos * kv_dim + i] = kv_store(s->k[i], s->kvbits);
			vb[(size_t)kv_pos * kv_dim + i] = kv_store(s->v[i], s->kvbits);
		}

		/* sink keys advance one position per step (fp16 round trip each time); :384-394 */
		for (int r = 0; r < kv_sink; r++) {
			for (int i = 0; i < kv_dim; i++) {
				s->k[i] = h2f(kb[(size_t)r * kv_dim + i]);
			}
			oracle_rope(s->k, kv_dim, p->head_dim, 1, p->rope_theta, p->rotary_dim);
			for (int i = 0; i < kv_dim; i++) {
				kb[(size_t)r * kv_dim + i] = kv_store(s->k[i], s->kvbits);
			}
		}

		/* :397-406 */
		int h;
#pragma omp parallel for private(h)
		for (h = 0; h < p->n_heads; h++) {
			oracle_attn_head(s->xb2 + h * p->head_dim, s->att + (size_t)h * p->seq_len, s->q + h * p->head_dim,
			                 kb + (h / kv_mul) * p->head_dim, vb + (h / kv_mul) * p->head_dim, p->head_dim, kv_dim, kv_len);
		}

		/* :410-415 */
		oracle_matvec(s->hb, s->xb2, w->wo[l], NULL, q_dim, dim, dbits);
		for (int i = 0; i < dim; i++) {
			x[i] += s->hb[i];
		}

		if (!p->norm_par) { /* :417-420 */
			oracle_norm(s->xb, x, w->rms_ffn_weight[l], dim, p->norm_eps, p->norm_ln);
		}

		float* moe_weights = s->exp + p->n_experts;
		int* moe_experts = (int*)moe_weights + nact;
		if (p->n_experts) { /* :425-432 */
			oracle_matvec(s->exp, s->xb, w->moegate[l], NULL, dim, p->n_experts, dbits);
			oracle_moe_gate(moe_weights, moe_experts, s->exp, p->n_experts, p->n_experts_ac);
			if (trace_experts) {
				memcpy(trace_experts + (size_t)l * nact, moe_experts, nact * sizeof(int));
				memcpy(trace_weights + (size_t)l * nact, moe_weights, nact * sizeof(float));
				memcpy(trace_logits + (size_t)l * p->n_experts, s->exp, p->n_experts * sizeof(float));
			}
		} else {
			moe_weights[0] = 1.0f;
			moe_experts[0] = 0;
		}

		/* :435-457 */
		for (int e = 0; e < nact; ++e) {
			size_t esize = (size_t)dim * hidden_dim * dbits / 8;
			const char* w1 = (const char*)w->w1[l] + moe_experts[e] * esize;
			const char* w2 = (const char*)w->w2[l] + moe_experts[e] * esize;
			const char* w3 = (const char*)w->w3[l] + moe_experts[e] * esize;
			oracle_matvec(s->hb, s->xb, w1, NULL, dim, hidden_dim, dbits);
			oracle_matvec(s->hb2, s->xb, w3, NULL, dim, hidden_dim, dbits);
			for (int i = 0; i < hidden_dim; i++) {
				s->hb[i] = oracle_act(s->hb[i], p->act_gelu) * s->hb2[i];
			}
			oracle_matvec(s->xb2, s->hb, w2, NULL, hidden_dim, dim, dbits);
			for (int i = 0; i < dim; i++) {
				x[i] += s->xb2[i] * moe_weights[e];
			}
		}
	}

	if ((flags & FF_UPDATE_KV_ONLY) || !(stage_flags & 2u)) { /* :460-463 */
		return NULL;
	}

	oracle_norm(x, x, w->rms_final_weight, dim, p->norm_eps, p->norm_ln);    /* :466 */
	oracle_matvec(s->logits, x, w->wcls, NULL, dim, p->vocab_size, dbits); /* :469 */
	return s->logits;
}

float* oracle_forward(struct Transformer* t, int token, int pos, unsigned flags) {
	return oracle_forward_stage(t, token, pos, flags, 3u);
}

/* xorshift* generator and the [0,1) coin; reference src/sampler.c:7-18 */
static unsigned int oracle_random_u32(unsigned long long* state) {
	unsigned long long s = *state;
	s ^= s >> 12;
	s ^= s << 25;
	s ^= s >> 27;
	*state = s;
	return (unsigned int)((s * 0x2545F4914F6CDD1Dull) >> 32);
}

/* min-p sampling; reference src/sampler.c:44-78.  Leaves `logits` alone (the reference overwrites it with the unscaled
 * probabilities, :56) -- the checker's callers compare logits afterwards. */
static int oracle_sample_minp(const float* logits, int n, float minp, float temperature, float coin) {
	float max_logit = -FLT_MAX;
	for (int i = 0; i < n; i++) {
		max_logit = logits[i] > max_logit ? logits[i] : max_logit;
	}
	float logit_cutoff = max_logit + logf(minp) * temperature; /* :52 */
	int fallback = 0;
	float cumulative_prob = 0.0f;
	for (int i = 0; i < n; i++) { /* :58-66 */
		if (logits[i] >= logit_cutoff) {
			cumulative_prob += expf((logits[i] - max_logit) / temperature);
			fallback = i;
		}
	}
	float r = coin * cumulative_prob; /* :69 */
	float cdf = 0.0f;
	for (int i = 0; i < n; i++) { /* :71-76 */
		if (logits[i] >= logit_cutoff) {
			cdf += expf((logits[i] - max_logit) / temperature);
			if (r < cdf) {
				return i;
			}
		}
	}
	return fallback;
}

int oracle_argmax(const float* logits, int n);

/* one draw of the host sampler; reference src/sampler.c:80-90.  *rng_state advances only when a coin is drawn. */
int oracle_sample(const float* logits, int n, float temperature, float minp, unsigned long long* rng_state) {
	if (temperature == 0.0f || minp >= 1.0f) {
		return oracle_argmax(logits, n);
	}
	float coin = (float)(oracle_random_u32(rng_state) >> 8) / 16777216.0f; /* :16-18 */
	return oracle_sample_minp(logits, n, minp, temperature, coin);
}

/* greedy sampler: first index of the strict maximum; reference src/sampler.c:34-42 */
int oracle_argmax(const float* logits, int n) {
	int max_i = -1;
	float max_p = -FLT_MAX;
	for (int i = 0; i < n; i++) {
		if (logits[i] > max_p) {
			max_i = i;
			max_p = logits[i];
		}
	}
	return max_i;
}
