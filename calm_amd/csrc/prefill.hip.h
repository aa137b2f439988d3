// prefill.hip.h -- batched prompt ingestion for the MI355X backend.
//
// The reference feeds a prompt one token at a time through forward(token, pos, FF_UPDATE_KV_ONLY)
// (src/run.c:208,216-218; "prompt processing is serial", README.md:80).  prefill_hip does the same work
// -- the KV cache rows of n consecutive positions -- PF_NT tokens at a time, so that every weight byte
// is streamed once per chunk instead of once per token, and the multiply-adds move to the matrix cores.
//
// Numerics stay those of the decode path: fp32 activations, exactly decoded weights, fp32 accumulation.
// v_mfma_f32_32x32x2_f32 takes f32 A and B operands and is bit-for-bit an fmaf chain (157 TF dense peak
// on this chip); the activations are NOT narrowed to fp16 / bf16 to reach the 16x faster MFMA forms --
// that costs 3e-4..2e-3 per matvec and breaks the 1e-3 logits parity with the CPU path.
//
// A workgroup (4 waves) owns one tile of 32 output units x 32 tokens; its waves split the reduction
// dimension (wave w takes every 4th step of 64 weights per row) and add their partial tiles through LDS in
// a fixed order.  Operands go global -> registers directly: A = 32 consecutive weights of this lane's row
// (lane = unit i, k-half kk), decoded to f32 in registers; B = the matching 32 activations of token j
// (lane = token j, k-half kk), contiguous in the token-major activation matrix (L2 resident: 64 x dim x 4 B).
// The k order inside the dot product is permuted (both operands agree), which fp32 addition does not mind
// beyond rounding.
#pragma once

namespace calm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PF_NT = 64; // tokens per chunk (two 32-token MFMA column tiles)

enum { PF_EPI_QKV = 0, PF_EPI_RESID = 1, PF_EPI_FFN_UP = 2 };

// embedding rows + RoPE table of a chunk of tokens   (src/infer.c:334-347, :223-236)
// grid = (ceil(max(dim, head_dim/2) / 256), PF_NT); rows of tokens b >= nb are zeroed
template <int DB>
__global__ void k_pf_begin(const int* tokens, int nb, int pos0, float* X, const void* embed, int dim, const float* rope_freq, float2* rope, int half_hd) {
	const int b = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < dim) {
		X[(size_t)b * dim + i] = b < nb ? decode_elem<DB>(embed, (size_t)tokens[b] * dim + i) : 0.f;
	}
	if (i < half_hd) {
		float val = (float)(pos0 + b) * rope_freq[i]; // src/infer.c:227-229
		rope[b * half_hd + i] = make_float2(cosf(val), sinf(val));
	}
}

// out[b][:] = norm(X[b][:]) * normw, one workgroup per token   (src/infer.c:183-207)
__global__ __launch_bounds__(256) void k_pf_norm(float* out, const float* X, const float* normw, int n, float eps, int ln) {
	__shared__ float red[16];
	const float* x = X + (size_t)blockIdx.x * n;
	float* o = out + (size_t)blockIdx.x * n;
	float mean = 0.f;
	if (ln) {
		float s = 0.f;
		for (int i = threadIdx.x; i < n; i += 256) {
			s += x[i];
		}
		mean = block_sum<256>(s, red) / (float)n;
	}
	float ss = 0.f;
	for (int i = threadIdx.x; i < n; i += 256) {
		float d = x[i] - mean;
		ss += d * d;
	}
	float var = block_sum<256>(ss, red) / (float)n;
	float scale = 1.0f / sqrtf(var + eps);
	for (int i = threadIdx.x; i < n; i += 256) {
		o[i] = (x[i] - mean) * scale * normw[i];
	}
}

struct PfGemmArgs {
	const float* xin;    // [PF_NT][K] activations, token-major
	const void *w0, *w1, *w2; // QKV: wq, wk, wv;  FFN_UP: w1, w3;  RESID: the matrix
	int K, M, nb;        // reduction length, output units, valid tokens
	float* out;          // QKV: Q [PF_NT][q_dim];  RESID: X [PF_NT][M] (accumulated into);  FFN_UP: H [PF_NT][M]
	const float* bqkv;
	const float2* rope;  // [PF_NT][head_dim / 2]
	void *kc, *vc;       // this layer's caches, [kv_head][seq_len][head_dim]
	int q_dim, kv_dim, head_dim, seq_len, kv_pos0;
	float clip;
	int gelu;
};

// one 16-byte piece of a weight row -> its G weights as f32 (exact in all three formats)
template <int DB>
__device__ __forceinline__ void pf_decode(u32x4 v, float (&wf)[Fmt<DB>::G]) {
	if constexpr (DB == 16) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			wf[2 * i] = half_bits_to_float((unsigned short)(v[i] & 0xffff));
			wf[2 * i + 1] = half_bits_to_float((unsigned short)(v[i] >> 16));
		}
	} else if constexpr (DB == 8) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			f32x2 lo = bf8x2_lo(v[i]), hi = bf8x2_hi(v[i]);
			wf[4 * i] = lo[0], wf[4 * i + 1] = lo[1], wf[4 * i + 2] = hi[0], wf[4 * i + 3] = hi[1];
		}
	} else {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			float s = bf8_byte0(v[i]) * -0.25f; // src/infer.c:37-40
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				int q = (int)((v[i] >> (8 + 3 * k)) & 7) - 4;
				wf[8 * i + k] = (float)q * s;
			}
		}
	}
}

// grid = (ceil(M / 32), ceil(nb / 32)), 256 threads
template <int DB, int KVB, int EPI>
__global__ __launch_bounds__(256) void k_pf_gemm(PfGemmArgs a) {
	constexpr int G = Fmt<DB>::G;
	constexpr int NMAT = EPI == PF_EPI_FFN_UP ? 2 : 1;
	constexpr int P = 32 / G; // 16-byte pieces of a row per lane and step
	__shared__ float part[3][NMAT * 16][64]; // partial tiles of waves 1..3

	const int lane = lane_id(), wave = wave_id();
	const int j = lane & 31, kk = lane >> 5;
	const int unit0 = blockIdx.x * 32, tok0 = blockIdx.y * 32;
	const size_t row_bytes = (size_t)a.K * DB / 8;
	const int npieces = a.K / G;                   // 16-byte pieces per row
	const int nsteps = (npieces + 2 * P - 1) / (2 * P); // a step = P pieces (32 weights) per k-half: 32 MFMAs per matrix

	// A: this lane's weight row(s); B: this lane's token row.  Indices are clamped, never branched on:
	// surplus lanes read real data and their results are dropped in the epilogue.
	const int u = min(unit0 + j, a.M - 1);
	const unsigned char* rowp[NMAT];
	if constexpr (EPI == PF_EPI_QKV) {
		const bool is_q = u < a.q_dim, is_k = u < a.q_dim + a.kv_dim;
		const unsigned char* base = (const unsigned char*)(is_q ? a.w0 : (is_k ? a.w1 : a.w2));
		const int ul = u - (is_q ? 0 : (is_k ? a.q_dim : a.q_dim + a.kv_dim));
		rowp[0] = base + (size_t)ul * row_bytes;
	} else if constexpr (EPI == PF_EPI_FFN_UP) {
		rowp[0] = (const unsigned char*)a.w0 + (size_t)u * row_bytes;
		rowp[1] = (const unsigned char*)a.w1 + (size_t)u * row_bytes;
	} else {
		rowp[0] = (const unsigned char*)a.w0 + (size_t)u * row_bytes;
	}
	const float* xrow = a.xin + (size_t)min(tok0 + j, a.nb - 1) * a.K;

	struct Frag {
		u32x4 w[NMAT][P];
		f32x4 x[8];
	};
	auto load = [&](Frag& f, int s) {
		const int p0 = (2 * min(s, nsteps - 1) + kk) * P;
#pragma unroll
		for (int i = 0; i < P; ++i) {
			const int piece = min(p0 + i, npieces - 1);
#pragma unroll
			for (int m = 0; m < NMAT; ++m) {
				f.w[m][i] = __builtin_nontemporal_load((gptr16)rowp[m] + piece);
			}
			const f32x4* xp = (const f32x4*)(xrow + (size_t)piece * G);
#pragma unroll
			for (int q = 0; q < G / 4; ++q) {
				f.x[i * (G / 4) + q] = xp[q];
			}
		}
	};

	f32x16 acc[NMAT];
#pragma unroll
	for (int m = 0; m < NMAT; ++m) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			acc[m][r] = 0.f;
		}
	}
	auto compute = [&](const Frag& f, int s) {
		const int p0 = (2 * s + kk) * P;
#pragma unroll
		for (int i = 0; i < P; ++i) {
			const bool valid = p0 + i < npieces; // ragged rows: pieces past the row's end multiply as zeros
			float wf[NMAT][G];
#pragma unroll
			for (int m = 0; m < NMAT; ++m) {
				u32x4 v = f.w[m][i];
				if (!valid) {
					v = (u32x4){0u, 0u, 0u, 0u}; // decodes to zeros in every format
				}
				pf_decode<DB>(v, wf[m]);
			}
#pragma unroll
			for (int e = 0; e < G; ++e) {
#pragma unroll
				for (int m = 0; m < NMAT; ++m) {
					acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[m][e], f.x[(i * G + e) / 4][(i * G + e) % 4], acc[m], 0, 0, 0);
				}
			}
		}
	};

	// wave w takes steps w, w+4, ...; two steps of operands stay in flight ahead of the one being multiplied.
	// The loop is unrolled by three so the three fragment buffers rotate by name (no register copies, which
	// would wait for the loads just issued), and a scheduling barrier after each load block keeps the
	// compiler from sinking the loads down to their first use (it did: 33 % of the f32 MFMA peak, vmcnt(0)
	// in front of every other MFMA).  Loads are clamped, never skipped, so s_waitcnt stays counted.
	Frag f0, f1, f2;
	load(f0, wave);
	load(f1, wave + 4);
	for (int s = wave; s < nsteps; s += 12) {
		load(f2, s + 8);
		__builtin_amdgcn_sched_barrier(0);
		compute(f0, s);
		load(f0, s + 12);
		__builtin_amdgcn_sched_barrier(0);
		if (s + 4 < nsteps) {
			compute(f1, s + 4);
		}
		load(f1, s + 16);
		__builtin_amdgcn_sched_barrier(0);
		if (s + 8 < nsteps) {
			compute(f2, s + 8);
		}
	}

	// add the four waves' partial tiles in wave order (deterministic)
	if (wave > 0) {
#pragma unroll
		for (int m = 0; m < NMAT; ++m) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				part[wave - 1][m * 16 + r][lane] = acc[m][r];
			}
		}
	}
	__syncthreads();
	if (wave > 0) {
		return;
	}
#pragma unroll
	for (int w = 0; w < 3; ++w) {
#pragma unroll
		for (int m = 0; m < NMAT; ++m) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				acc[m][r] += part[w][m * 16 + r][lane];
			}
		}
	}

	// C layout: column (token) = lane & 31, row (unit) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
	const int token = tok0 + j;
	if (token >= a.nb) {
		return;
	}
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int ub = unit0 + 8 * g + 4 * kk; // four consecutive units; M is a multiple of 4
		if (ub >= a.M) {
			continue;
		}
		if constexpr (EPI == PF_EPI_RESID) {
			float4* p = (float4*)(a.out + (size_t)token * a.M + ub);
			float4 t = *p;
			t.x += acc[0][4 * g], t.y += acc[0][4 * g + 1], t.z += acc[0][4 * g + 2], t.w += acc[0][4 * g + 3];
			*p = t;
		} else if constexpr (EPI == PF_EPI_FFN_UP) {
			float h[4];
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				float up = acc[0][4 * g + e], gt = acc[1][4 * g + e];
				h[e] = (a.gelu ? act_gelu(up) : act_silu(up)) * gt; // src/infer.c:440-450
			}
			*(float4*)(a.out + (size_t)token * a.M + ub) = make_float4(h[0], h[1], h[2], h[3]);
		} else {
#pragma unroll
			for (int pr = 0; pr < 2; ++pr) { // RoPE pairs (2i, 2i+1); q / k / v boundaries are multiples of 8
				const int uu = ub + 2 * pr;
				float v0 = acc[0][4 * g + 2 * pr], v1 = acc[0][4 * g + 2 * pr + 1];
				if (a.bqkv) {
					v0 += a.bqkv[uu];
					v1 += a.bqkv[uu + 1];
				}
				v0 = clipf(v0, a.clip);
				v1 = clipf(v1, a.clip);
				if (uu < a.q_dim + a.kv_dim) { // src/infer.c:223-236
					const int ul = uu < a.q_dim ? uu : uu - a.q_dim;
					const float2 cs = a.rope[(size_t)token * (a.head_dim >> 1) + ((ul % a.head_dim) >> 1)];
					const float r0 = v0 * cs.x - v1 * cs.y, r1 = v0 * cs.y + v1 * cs.x;
					v0 = r0, v1 = r1;
				}
				if (uu < a.q_dim) {
					*(float2*)(a.out + (size_t)token * a.q_dim + uu) = make_float2(v0, v1);
				} else {
					int jl = uu - a.q_dim;
					void* cache = a.kc;
					if (jl >= a.kv_dim) {
						jl -= a.kv_dim;
						cache = a.vc;
					}
					const size_t off = ((size_t)(jl / a.head_dim) * a.seq_len + a.kv_pos0 + token) * a.head_dim + (jl % a.head_dim);
					if constexpr (KVB == 16) {
						*(__half2*)((__half*)cache + off) = __floats2half2_rn(v0, v1); // src/infer.c:378-381
					} else {
						*(unsigned short*)((unsigned char*)cache + off) = (unsigned short)(__builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, 0, false) & 0xffff);
					}
				}
			}
		}
	}
}

} // namespace calm
