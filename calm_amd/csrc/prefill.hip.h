// prefill.hip.h -- batched prompt ingestion for the MI355X backend.
//
// The reference feeds a prompt one token at a time through forward(token, pos, FF_UPDATE_KV_ONLY)
// (src/run.c:208,216-218; "prompt processing is serial", README.md:80).  prefill_hip does the same work
// -- the KV cache rows of n consecutive positions -- up to PF_NT_DENSE (mixture-of-experts models: PF_NT) tokens at a time, so that every weight byte
// is streamed once per chunk instead of once per token, and the multiply-adds move to the matrix cores.
//
// Numerics: exactly decoded weights (all three formats are exact in binary16), fp32 accumulation, and the fp32
// activations as the SUM OF TWO binary16 numbers, x = hi + lo with hi = half(x), lo = half(x - hi): 22 significand
// bits, |x - (hi + lo)| <= 2^-22 |x| (2^-25 absolute below 2^-14).  Each weight x activation product is then two
// v_mfma_f32_32x32x16_f16 products, both exact in fp32 -- 16x the rate of the f32 MFMA form for twice the
// instructions.  Narrowing the activations to ONE fp16 / bf16 number costs 3e-4..2e-3 per matvec and breaks the
// 1e-3 logits parity with the CPU path; the split measures 4..8e-7 (tools/hilo_study.py), the same as the f32
// MFMA form this file used before (bit-for-bit an fmaf chain, 157 TF peak).  An activation beyond +-65504 (or a NaN) raises
// the model's range flag (calm_pf_range_ptr) and the prompt is redone by the serial fp32 path (infer_hip.hip: prefill_impl).
//
// Two GEMM forms.  k_pf_gemm (next paragraph) fills the chip from few tiles by splitting K over the waves of a workgroup: short
// prompts.  k_pf_gemm_wide (further down: B staged once per workgroup through LDS, A through a wave-private LDS image, XCD-aware
// workgroup order, optionally K cut into ranges across workgroups) moves 2.5 x fewer operand bytes per multiply-add through the
// vector memory path and takes over as soon as its 256 x 64 tiles cover the chip.  Attention has its own matrix-core kernel.
//
// GEMM structure (k_pf_gemm).  A wave owns NA x 2 accumulator tiles of 32 units x 32 tokens: NA = 1..3 unit
// strips (QKV, residual, classifier GEMMs; the count that wastes the fewest workgroup rounds) or the w1 / w3 pair
// of one strip (FFN-up), times two token tiles -- so every decoded weight feeds two MFMAs and every activation
// fragment feeds NA, which cuts the operand traffic per MFMA (one strip x one token tile measured 40 % of the
// MFMA peak with the operand fetch in the way).  The four waves of a workgroup split the reduction dimension
// (wave w takes every 4th sub-step of a row) and add their partial tiles through LDS in a fixed order.
// Operands go global -> registers:
//   A  consecutive weights of this lane's row (lane = unit i, k-half kk), 8 per MFMA, decoded to binary16;
//   B  the matching activations of token j (lane = token j, k-half kk), hi and lo, from a FRAGMENT-MAJOR activation
//      matrix (pf_unit below): a wave's 16-byte loads are 1 KiB contiguous, and the matrix is L2 resident.
// The k order inside the dot product is permuted (both operands agree), which fp32 addition does not mind
// beyond rounding.  Operands of the next sub-step are loaded (double buffer, scheduling barrier) before the
// current one's 8..48 MFMAs are issued.  Mixture-of-experts layers run the same kernel as a grouped GEMM
// (k_pf_route packs the rows, a workgroup column looks up its expert).
#pragma once

namespace calm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int PF_NT = 1024; // the least a dense model's prompt chunk is (and k_pf_route's workgroup; mixture-of-experts chunks: PF_NT_MOE)
// Dense models take chunks of up to PF_NT_DENSE tokens (knob "pf_chunk"): at 1024 tokens the grids of the QKV / wo / w2 GEMMs cover 256-384
// of the 512 workgroup slots, at 2048 all of them -- + 4...11 % per GEMM (tools/experiments/exp_pfgemm_big.hip, profiles/r04_prefill.txt)
constexpr int PF_NT_DENSE = 2048;

enum { PF_EPI_QKV = 0, PF_EPI_RESID = 1, PF_EPI_FFN_UP = 2, PF_EPI_STORE = 3 };
constexpr int PF_MAX_ACTIVE = 8; // experts per token the batched MoE routing handles

// Fragment-major activation matrix of rows of n values (GEMM B operand), every value as two binary16 numbers.
// The 16-byte unit holding the hi halves of columns k..k+7 (k % 8 == 0) of token t lives at unit index
//     (((t / 32 * nsteps + k / 64) * 4 + k % 32 / 8) * 2 + 0) * 64 + (k % 64 / 32) * 32 + t % 32,   nsteps = ceil(n / 64)
// and the unit with their lo halves 64 units further: [token group of 32][step of 64 columns][MFMA m of the lane's 32
// columns][hi, lo][lane = (k-half, token)] -- exactly the order in which a wave's lanes consume it, 8 KiB per (group,
// step) like the fp32 values it stands for.  Rows are padded to whole steps; the padding is never written and stays
// zero from the allocation.
__device__ __forceinline__ size_t pf_unit(int t, int k, int nsteps) {
	return ((((size_t)(t >> 5) * nsteps + (k >> 6)) * 4 + ((k & 31) >> 3)) << 7) + (((k >> 5) & 1) << 5) + (t & 31);
}
// Raised (never cleared by a kernel) when an activation did not fit the hi + lo form: beyond the binary16 range, or not a number.
// The word lives in pinned host memory of the model being ingested, one word per chunk of the call (Ctx::pf_flag; prefill_impl points
// this device's copy of calm_pf_range_ptr at the chunk's word on the stream ahead of the chunk's kernels) and is read by the host at
// the call's own final synchronisation -- no extra round trip per chunk, no flag shared between models: a prompt during which one was
// raised is redone from that chunk on through the serial fp32 decode path, so the batched path either agrees with the serial one to
// fp32 rounding or is not used.
__device__ unsigned* calm_pf_range_ptr;

// x = hi + lo, two binary16 numbers each.  Out-of-range values saturate HERE (NaN stays NaN: the compares are ordered) and
// raise the model's range flag (calm_pf_range_ptr), which sends the prompt back through the serial path.
__device__ __forceinline__ void pf_split2(float a, float b, unsigned& hi, unsigned& lo) {
	if (!(fabsf(a) <= 65504.f) || !(fabsf(b) <= 65504.f)) { // also true for NaN
		if (calm_pf_range_ptr) { // (null only for kernels launched outside prefill_impl: the unit-test hooks)
			*calm_pf_range_ptr = 1u;
		}
	}
	a = a > 65504.f ? 65504.f : (a < -65504.f ? -65504.f : a);
	b = b > 65504.f ? 65504.f : (b < -65504.f ? -65504.f : b);
	const __half2 h = __floats2half2_rn(a, b);
	const float2 hf = __half22float2(h);
	const __half2 l = __floats2half2_rn(a - hf.x, b - hf.y);
	hi = __builtin_bit_cast(unsigned, h);
	lo = __builtin_bit_cast(unsigned, l);
}
// columns k..k+7 of token t <- v[0..7]
__device__ __forceinline__ void pf_store8(void* m, int t, int k, int nsteps, const float (&v)[8]) {
	u32x4 hi, lo;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		unsigned h, l;
		pf_split2(v[2 * i], v[2 * i + 1], h, l);
		hi[i] = h, lo[i] = l;
	}
	u32x4* u = (u32x4*)m + pf_unit(t, k, nsteps);
	u[0] = hi;
	u[64] = lo;
}
// columns k..k+3 (k % 4 == 0) of token t <- v[0..3]: half a unit
__device__ __forceinline__ void pf_store4(void* m, int t, int k, int nsteps, const float (&v)[4]) {
	unsigned h0, l0, h1, l1;
	pf_split2(v[0], v[1], h0, l0);
	pf_split2(v[2], v[3], h1, l1);
	const u32x2 hi = {h0, h1}, lo = {l0, l1};
	u32x2* u = (u32x2*)((u32x4*)m + pf_unit(t, k & ~7, nsteps)) + ((k >> 2) & 1);
	u[0] = hi;
	u[128] = lo;
}
__host__ __device__ inline int pf_steps(int n) {
	return (n + 63) >> 6;
}

// embedding rows + RoPE table of a chunk of tokens   (src/infer.c:334-347, :223-236)
// grid = (ceil(max(dim, head_dim/2) / 256), nb)
template <int DB>
__global__ void k_pf_begin(const int* tokens, int pos0, float* X, const void* embed, int dim, const float* rope_freq, float2* rope, int half_hd) {
	const int b = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (embed && i < dim) { // (a later pipeline stage receives X from the stage before it)
		X[(size_t)b * dim + i] = decode_elem<DB>(embed, (size_t)tokens[b] * dim + i);
	}
	if (i < half_hd) {
		float val = (float)(pos0 + b) * rope_freq[i]; // src/infer.c:227-229
		rope[b * half_hd + i] = make_float2(cosf(val), sinf(val));
	}
}

// out (fragment-major) = norm(X[t][:]) * normw   (src/infer.c:183-207)
// A workgroup takes EIGHT consecutive tokens (blockIdx.x) and one of gridDim.y column slices: a 128-byte line of the fragment-major
// matrix is the same 8 columns of 8 consecutive tokens (pf_unit: the token is the lane), so one workgroup writes whole lines -- with
// one token per workgroup (rounds 1-5) every line was put together in the L2 out of eight workgroups' 16-byte pieces: 34.5 us for
// the 2048 x 4096 matrix, 67 MB moved at 1.9 TB/s.  Pass 1: the rows' sums of squares, all threads on all eight rows (whole rows,
// whatever the slice: the column slices of a token group recompute them -- reads out of the L2); pass 2: thread = (token, block of 8 columns).
// grid = (ceil(nb / 8), slices), 256 threads; n % 32 == 0
__global__ __launch_bounds__(256) void k_pf_norm(void* out, const float* X, const float* normw, int n, float eps, int ln, int nb) {
	__shared__ float sm_part[2][8][4], sm_mean[8];
	const int t0 = blockIdx.x * 8, lane = lane_id(), wave = wave_id();
	const int n4 = n >> 2;
	// pass 1: all 256 threads on each of the eight rows (eight loads in flight per thread), the waves' sums through LDS
	const float4* rows[8];
#pragma unroll
	for (int tl = 0; tl < 8; ++tl) {
		rows[tl] = (const float4*)(X + (size_t)min(t0 + tl, nb - 1) * n);
	}
	float mean8[8];
#pragma unroll
	for (int tl = 0; tl < 8; ++tl) {
		mean8[tl] = 0.f;
	}
	if (ln) {
		float s8[8];
#pragma unroll
		for (int tl = 0; tl < 8; ++tl) {
			s8[tl] = 0.f;
		}
		for (int i = threadIdx.x; i < n4; i += 256) {
#pragma unroll
			for (int tl = 0; tl < 8; ++tl) {
				const float4 v = rows[tl][i];
				s8[tl] += (v.x + v.y) + (v.z + v.w);
			}
		}
#pragma unroll
		for (int tl = 0; tl < 8; ++tl) {
			const float s = wave_sum(s8[tl]);
			if (lane == 0) {
				sm_part[0][tl][wave] = s;
			}
		}
		__syncthreads();
#pragma unroll
		for (int tl = 0; tl < 8; ++tl) {
			mean8[tl] = ((sm_part[0][tl][0] + sm_part[0][tl][1]) + (sm_part[0][tl][2] + sm_part[0][tl][3])) / (float)n;
		}
	}
	{
		float s8[8];
#pragma unroll
		for (int tl = 0; tl < 8; ++tl) {
			s8[tl] = 0.f;
		}
		for (int i = threadIdx.x; i < n4; i += 256) {
#pragma unroll
			for (int tl = 0; tl < 8; ++tl) {
				const float4 v = rows[tl][i];
				const float a = v.x - mean8[tl], b = v.y - mean8[tl], c = v.z - mean8[tl], d = v.w - mean8[tl];
				s8[tl] += (a * a + b * b) + (c * c + d * d);
			}
		}
#pragma unroll
		for (int tl = 0; tl < 8; ++tl) {
			const float s = wave_sum(s8[tl]);
			if (lane == 0) {
				sm_part[1][tl][wave] = s;
			}
		}
		if (threadIdx.x == 0) { // (every thread holds all eight means)
#pragma unroll
			for (int tl = 0; tl < 8; ++tl) {
				sm_mean[tl] = mean8[tl];
			}
		}
	}
	__syncthreads();
	const int tl = threadIdx.x & 7, t = t0 + tl;
	const float var = ((sm_part[1][tl][0] + sm_part[1][tl][1]) + (sm_part[1][tl][2] + sm_part[1][tl][3])) / (float)n;
	const float mean = sm_mean[tl], scale = 1.0f / sqrtf(var + eps);
	const float4* x4 = (const float4*)(X + (size_t)min(t, nb - 1) * n);
	const float4* w4 = (const float4*)normw;
	const int nsteps = pf_steps(n), nkb = n >> 3;
	const int per = (nkb + (int)gridDim.y - 1) / (int)gridDim.y;
	const int kb1 = min(nkb, ((int)blockIdx.y + 1) * per);
	for (int kb = (int)blockIdx.y * per + (int)(threadIdx.x >> 3); kb < kb1; kb += 32) {
		float v[8];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const float4 a = x4[2 * kb + h], g = w4[2 * kb + h];
			v[4 * h] = (a.x - mean) * scale * g.x;
			v[4 * h + 1] = (a.y - mean) * scale * g.y;
			v[4 * h + 2] = (a.z - mean) * scale * g.z;
			v[4 * h + 3] = (a.w - mean) * scale * g.w;
		}
		if (t < nb) {
			pf_store8(out, t, 8 * kb, nsteps, v);
		}
	}
}
// its grid: 8-token groups x as many column slices as bring the grid to two workgroups per CU (at most 8, at least 64 columns each)
inline dim3 pf_norm_grid(int nb, int n, int ncu) {
	const int groups = (nb + 7) / 8;
	int slices = (2 * ncu + groups - 1) / groups;
	slices = slices > 8 ? 8 : slices;
	while (slices > 1 && (n >> 3) / slices < 8) {
		--slices;
	}
	return dim3(groups, slices < 1 ? 1 : slices);
}

// Mixture-of-experts routing of a chunk (src/infer.c:277-305 per token): top-k by logit, first maximum wins
// ties, weights = softmax over the selected logits.  The (token, rank) pairs are then packed expert by expert
// into rows of ONE matrix, every expert's group padded to whole workgroup columns -- `gran` 64-row columns at a time: 1 for the
// K-split / wide GEMM forms, 2 where the grouped GEMMs take the big form's 128-token tiles -- so that a single grouped GEMM launch
// serves all experts:
//   rows[r]      token of packed row r, or -1 for padding          col_expert[c]  expert of 64-row column c, -1 past the end
//   slot[t*k+j]  packed row of token t's rank-j expert             wsel[t*k+j]    its routing weight
// One workgroup of PF_NT threads, thread t = tokens t, t + PF_NT, ... (chunks of up to PF_NT_MOE = 4 PF_NT tokens); everything is in
// token order (deterministic): a token's place in its expert's group is the number of earlier tokens routed to that expert -- a
// ballot per expert within the wave plus the counts of the (pass, wave) pairs before it (a token routes to an expert at most once).
constexpr int PF_ROUTE_TPT = 4;                 // tokens per thread of k_pf_route
constexpr int PF_NT_MOE = PF_ROUTE_TPT * PF_NT; // the most a mixture-of-experts model's chunk is (knob "pf_chunk_moe", which it is the default of)
__global__ __launch_bounds__(PF_NT) void k_pf_route(const float* gate, int nb, int n_experts, int n_active, int max_cols, int gran, int* rows, int* col_expert, int* slot,
                                                    float* wsel) {
	constexpr int NW = PF_NT / 64, TPT = PF_ROUTE_TPT;
	__shared__ int wave_cnt[CALM_MAX_EXPERTS][TPT * NW]; // tokens of (pass, wave) routed to expert e, then the exclusive prefix over them
	__shared__ int first_col[CALM_MAX_EXPERTS + 1];
	__shared__ int cnt[CALM_MAX_EXPERTS];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int mine[TPT][PF_MAX_ACTIVE], inwave[TPT][PF_MAX_ACTIVE];
#pragma unroll
	for (int p = 0; p < TPT; ++p) {
#pragma unroll
		for (int k = 0; k < PF_MAX_ACTIVE; ++k) {
			mine[p][k] = -1, inwave[p][k] = 0;
		}
	}
#pragma unroll
	for (int p = 0; p < TPT; ++p) {
		const int t = p * PF_NT + threadIdx.x;
		if (t < nb) {
			const float* g = gate + (size_t)t * n_experts;
			float max_val = -3.402823466e+38f;
			for (int j = 0; j < n_experts; ++j) {
				max_val = max_val < g[j] ? g[j] : max_val;
			}
			unsigned long long mask = 0;
			float wsum = 0.f;
#pragma unroll
			for (int k = 0; k < PF_MAX_ACTIVE; ++k) {
				if (k < n_active) {
					int best = -1;
					for (int j = 0; j < n_experts; ++j) {
						if ((mask & (1ull << j)) == 0 && (best == -1 || g[j] > g[best])) {
							best = j;
						}
					}
					mine[p][k] = best;
					wsum += expf(g[best] - max_val);
					mask |= 1ull << best;
				}
			}
#pragma unroll
			for (int k = 0; k < PF_MAX_ACTIVE; ++k) {
				if (k < n_active) {
					wsel[t * n_active + k] = expf(g[mine[p][k]] - max_val) / wsum;
				}
			}
		}
	}
	const unsigned long long below = (1ull << lane) - 1;
	for (int e = 0; e < n_experts; ++e) {
#pragma unroll
		for (int p = 0; p < TPT; ++p) {
			bool to_e = false;
#pragma unroll
			for (int k = 0; k < PF_MAX_ACTIVE; ++k) {
				to_e |= mine[p][k] == e;
			}
			const unsigned long long m = __ballot(to_e);
			if (lane == 0) {
				wave_cnt[e][p * NW + wave] = __popcll(m);
			}
#pragma unroll
			for (int k = 0; k < PF_MAX_ACTIVE; ++k) {
				if (mine[p][k] == e) {
					inwave[p][k] = __popcll(m & below);
				}
			}
		}
	}
	__syncthreads();
	if ((int)threadIdx.x < n_experts) {
		int c = 0;
		for (int w = 0; w < TPT * NW; ++w) { // (pass-major: tokens 0 .. PF_NT - 1 come before PF_NT ..)
			const int n = wave_cnt[threadIdx.x][w];
			wave_cnt[threadIdx.x][w] = c;
			c += n;
		}
		cnt[threadIdx.x] = c;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int col = 0;
		for (int e = 0; e < n_experts; ++e) {
			first_col[e] = col;
			const int ncols = (cnt[e] + 64 * gran - 1) / (64 * gran) * gran;
			for (int c = 0; c < ncols; ++c) {
				col_expert[col++] = e;
			}
		}
		first_col[n_experts] = col;
		for (; col < max_cols; ++col) {
			col_expert[col] = -1;
		}
	}
	__syncthreads();
#pragma unroll
	for (int p = 0; p < TPT; ++p) {
		const int t = p * PF_NT + threadIdx.x;
#pragma unroll
		for (int k = 0; k < PF_MAX_ACTIVE; ++k) {
			if (mine[p][k] >= 0) {
				const int r = first_col[mine[p][k]] * 64 + wave_cnt[mine[p][k]][p * NW + wave] + inwave[p][k];
				rows[r] = t;
				slot[t * n_active + k] = r;
			}
		}
	}
	// the padding of every expert's last column(s)
	for (int e = 0; e < n_experts; ++e) {
		for (int r = first_col[e] * 64 + cnt[e] + threadIdx.x; r < first_col[e + 1] * 64; r += PF_NT) {
			rows[r] = -1;
		}
	}
}

// packed row r (fragment-major) = row rows[r] of src; padding rows (rows[r] < 0) are zero-filled.
// A wave writes whole 1 KiB blocks of the destination ([32-row group][step][MFMA][hi, lo][lane]: 64 consecutive units) and
// fetches, per lane, the unit of its row's source token from the L2-resident source matrix.
// grid = (row groups of 32, ceil(nsteps / 8)), 256 threads
__global__ __launch_bounds__(256) void k_pf_gather(float4* dst, const float4* src, const int* rows, const int* col_expert, int n) {
	const int g = blockIdx.x, lane = lane_id(), wave = wave_id();
	if (col_expert[g >> 1] < 0) {
		return;
	}
	const int t = rows[32 * g + (lane & 31)], nsteps = pf_steps(n);
	const int s0 = blockIdx.y * 8, s1 = min(s0 + 8, nsteps);
	const int ts = t < 0 ? 0 : t; // a padding row is written as zeros: whatever a longer chunk left there went through the FFN-up's
	                              // epilogue and could raise the range flag (a serial redo for nothing)
	const float4* sp = src + (size_t)(ts >> 5) * nsteps * 512 + (lane & 32) + (ts & 31);
	float4* dp = dst + (size_t)g * nsteps * 512 + lane;
	for (int b = s0 * 8 + wave; b < s1 * 8; b += 4) { // block b: step b / 8, (MFMA, hi / lo) b % 8
		const float4 v = sp[(size_t)b * 64];
		dp[(size_t)b * 64] = t < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : v;
	}
}

// x[t] += sum_j wsel[t][j] * y[slot[t][j]]: the experts' outputs added in rank order (src/infer.c:452-456)
__global__ __launch_bounds__(256) void k_pf_combine(float* X, const float* Y, const int* slot, const float* wsel, int n_active, int dim) {
	const int t = blockIdx.x;
	for (int i = threadIdx.x; i < (dim >> 2); i += 256) {
		float4 x = ((float4*)(X + (size_t)t * dim))[i];
		for (int j = 0; j < n_active; ++j) {
			const float w = wsel[t * n_active + j];
			const float4 y = ((const float4*)(Y + (size_t)slot[t * n_active + j] * dim))[i];
			x.x += w * y.x, x.y += w * y.y, x.z += w * y.z, x.w += w * y.w;
		}
		((float4*)(X + (size_t)t * dim))[i] = x;
	}
}

// ---- causal attention of a chunk of tokens over the cache   (src/infer.c:238-267,397-406 per token) ----------
// The decode kernel k_attn with a query dimension: one workgroup per (head, group of TQ consecutive tokens).
// Every cached K / V row is loaded once and used for all TQ queries (per-token launches of k_attn were bound
// by L2 bandwidth: 256 tokens x 32 heads each streaming the whole context).  Query b attends to rows
// [0, pf_kv0 + b]; rows past a query's own position are masked.  Output: fragment-major rows (pf_unit).
constexpr int PF_ATTN_BLOCK = 512; // 8 waves: 256 VGPRs per lane for the TQ query states (16 waves spill)
template <int LPR>
struct PfAttn {
	static constexpr int TQ = LPR <= 16 ? 4 : (LPR == 32 ? 2 : 1); // bounded by the LDS merge buffer (TQ x 8 waves x head_dim floats)
};

template <int KVB, int LPR>
__global__ __launch_bounds__(PF_ATTN_BLOCK) void k_pf_attn(AttnArgs a) {
	constexpr int RPW = 64 / LPR; // positions per wave-load
	constexpr int NW = PF_ATTN_BLOCK / 64;
	constexpr int UA = 4; // tiles in flight per wave
	constexpr int TQ = PfAttn<LPR>::TQ;
	__shared__ float sm_m[TQ][NW], sm_l[TQ][NW];
	__shared__ float sm_o[TQ][NW][LPR * 8];

	const int lane = lane_id(), wave = wave_id();
	const int h = blockIdx.x, b0 = blockIdx.y * TQ;
	const int kvh = h / a.kv_mul;
	const int r = lane % LPR, g = lane / LPR;
	const bool dvalid = r * 8 < a.head_dim;
	const int d0 = dvalid ? r * 8 : 0;
	const int nq = min(TQ, a.pf_nb - b0);      // queries of this group that exist
	const int kv_max = a.pf_kv0 + b0 + nq;     // rows the last of them attends to

	float qv[TQ][8];
#pragma unroll
	for (int q = 0; q < TQ; ++q) {
		const float* qsrc = a.q + (size_t)min(b0 + q, a.pf_nb - 1) * a.pf_stride + h * a.head_dim + d0;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			float qi = qsrc[i];
			qv[q][i] = dvalid ? qi : 0.f;
		}
	}
	const float inv_sqrt_hd = 1.0f / sqrtf((float)a.head_dim); // one rounding away from the reference's division (src/infer.c:247)

	float m[TQ], l[TQ], o[TQ][8];
#pragma unroll
	for (int q = 0; q < TQ; ++q) {
		m[q] = -INFINITY, l[q] = 0.f;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			o[q][i] = 0.f;
		}
	}

	constexpr int EB = KVB / 8;
	const unsigned char* kbase = (const unsigned char*)a.kc + ((size_t)kvh * a.seq_len * a.head_dim + d0) * EB;
	const unsigned char* vbase = (const unsigned char*)a.vc + ((size_t)kvh * a.seq_len * a.head_dim + d0) * EB;
	const size_t rstride = (size_t)a.head_dim * EB;

	for (int tb = wave * RPW; tb < kv_max; tb += NW * RPW * UA) {
		float kf[UA][8], vf[UA][8];
		int trow[UA];
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			trow[u] = tb + u * NW * RPW + g;
			const int t = min(trow[u], kv_max - 1); // always load (clamped); masked below
			if constexpr (KVB == 16) {
				u32x4 kw = *(const u32x4*)(kbase + (size_t)t * rstride);
				u32x4 vw = *(const u32x4*)(vbase + (size_t)t * rstride);
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					kf[u][2 * i] = half_bits_to_float((unsigned short)(kw[i] & 0xffff));
					kf[u][2 * i + 1] = half_bits_to_float((unsigned short)(kw[i] >> 16));
					vf[u][2 * i] = half_bits_to_float((unsigned short)(vw[i] & 0xffff));
					vf[u][2 * i + 1] = half_bits_to_float((unsigned short)(vw[i] >> 16));
				}
			} else {
				u32x2 kw = *(const u32x2*)(kbase + (size_t)t * rstride);
				u32x2 vw = *(const u32x2*)(vbase + (size_t)t * rstride);
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					f32x2 k0 = bf8x2_lo(kw[i]), k1 = bf8x2_hi(kw[i]);
					f32x2 v0 = bf8x2_lo(vw[i]), v1 = bf8x2_hi(vw[i]);
					kf[u][4 * i] = k0[0], kf[u][4 * i + 1] = k0[1], kf[u][4 * i + 2] = k1[0], kf[u][4 * i + 3] = k1[1];
					vf[u][4 * i] = v0[0], vf[u][4 * i + 1] = v0[1], vf[u][4 * i + 2] = v1[0], vf[u][4 * i + 3] = v1[1];
				}
			}
		}
#pragma unroll
		for (int q = 0; q < TQ; ++q) {
			const int kv_len_q = a.pf_kv0 + b0 + q + 1; // src/infer.c:332 for this token
			float s[UA];
#pragma unroll
			for (int u = 0; u < UA; ++u) {
				float d = 0.f;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					d = fmaf(qv[q][i], kf[u][i], d);
				}
				d = group_sum<LPR>(d);
				s[u] = trow[u] < kv_len_q ? d * inv_sqrt_hd : -INFINITY;
			}
			float mn = m[q];
#pragma unroll
			for (int u = 0; u < UA; ++u) {
				mn = fmaxf(mn, s[u]);
			}
			if (mn != -INFINITY) {
				float c = (m[q] == -INFINITY) ? 0.f : __expf(m[q] - mn);
				l[q] *= c;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					o[q][i] *= c;
				}
#pragma unroll
				for (int u = 0; u < UA; ++u) {
					float p = s[u] == -INFINITY ? 0.f : __expf(s[u] - mn);
					l[q] += p;
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						o[q][i] = fmaf(p, vf[u][i], o[q][i]);
					}
				}
				m[q] = mn;
			}
		}
	}

	// merge the lane groups of the wave, then the waves through LDS; wave q finishes query q
#pragma unroll
	for (int q = 0; q < TQ; ++q) {
#pragma unroll
		for (int ofs = LPR; ofs < 64; ofs <<= 1) {
			float m2 = __shfl_xor(m[q], ofs), l2 = __shfl_xor(l[q], ofs), o2[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				o2[i] = __shfl_xor(o[q][i], ofs);
			}
			sm_merge(m[q], l[q], o[q], m2, l2, o2);
		}
		if (g == 0) {
			if (r == 0) {
				sm_m[q][wave] = m[q];
				sm_l[q][wave] = l[q];
			}
			if (dvalid) {
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					sm_o[q][wave][d0 + i] = o[q][i];
				}
			}
		}
	}
	__syncthreads();
	if (wave < nq && g == 0) {
		const int q = wave;
		float mm = sm_m[q][0], ll = sm_l[q][0], oo[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			oo[i] = sm_o[q][0][d0 + i];
		}
#pragma unroll
		for (int w = 1; w < NW; ++w) {
			float o2[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				o2[i] = sm_o[q][w][d0 + i];
			}
			sm_merge(mm, ll, oo, sm_m[q][w], sm_l[q][w], o2);
		}
		if (dvalid) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				oo[i] /= ll;
			}
			pf_store8(a.out, b0 + q, h * a.head_dim + d0, pf_steps(a.pf_stride), oo);
		}
	}
}

// ---- the same attention on the matrix cores ------------------------------------------------------------------------------
// k_pf_attn spends its time in lane arithmetic (one dot-product lane group per cached row: 519 us for a 1024-token chunk of the
// Mistral-7B shape, as long as all four weight GEMMs of the layer once those run on the f16 MFMA).  Here a wave owns one query
// head and a tile of 32 consecutive tokens; the workgroup's four waves are the query heads that share a kv head (or, with
// fewer than four of them, further token tiles), so the K / V rows of a 32-key tile are fetched once per workgroup into LDS:
//   S^T[key][query] = K[key][:] . q[query][:]      A = K rows (binary16 as cached; e5m2 widened), B = q as hi + lo binary16
//   online softmax down the columns: a lane holds 16 keys of ONE query (C layout: column = lane & 31), so the running
//   maximum and sum are in-lane reductions plus one exchange with lane ^ 32
//   O^T[d][query] += V^T[d][key] . P[key][query]   A = V transposed (by the LDS write), B = P as hi + lo binary16, taken from
//   S^T's accumulator registers as they are: a lane's 8 registers are the 8 keys the MFMA wants from it, in the order
//   key = 16 u + 8 (e >> 2) + 4 (lane >> 5) + (e & 3), and the V^T image is written in that key order
// Every product is exact in fp32 (binary16 x binary16), accumulation is fp32: the result agrees with k_pf_attn to fp32 rounding.
// LDS images (per 32-key tile, ring of 3): K [key][HD] with the 16-byte chunk index XOR-swizzled by the key, V^T [d][32 keys]
// (64 B rows) with the 8-key chunk index XOR-swizzled by (d >> 3) & 3 -- both make the ds_read_b128 operand fetches conflict
// free for the hardware's 16-lane groups.
// VT (head size 128, where the backend keeps the value cache a second time, transposed in blocks -- kernels.hip.h attn_vt_offset):
// a.vc is that transposed cache, the V^T image is a plain copy of it (one 16-byte store per lane and pass instead of eight 2-byte
// ones), and the K rows take the permutation instead: A row r of the S^T product is key (r with bits 2 and 3 exchanged) of the tile,
// so that the 8 scores a lane contributes to k-step u are the 8 consecutive keys 16 u + 8 (lane >> 5) .. + 7.
// grid = (n_kv_heads, ceil(nb / (32 * TW))), 256 threads;  HG = heads per round (4, 2 or 1, dividing kv_mul), TW = 4 / HG
template <int KVB, int HD, int HG, bool VT>
__global__ __launch_bounds__(256, 2) void k_pf_attn_mfma(AttnArgs a) {
	static_assert(!VT || HD == 128, "the transposed value cache exists for head size 128");
	constexpr int TW = 4 / HG;       // token tiles per workgroup
	constexpr int NT = HD / 16;      // MFMA k-steps of a q . k dot product
	constexpr int ND = HD / 32;      // 32-row tiles of O^T
	constexpr int CK = HD / 8;       // 16-byte chunks of a K row
	__shared__ u32x4 kst[3][32 * CK];
	__shared__ u32x4 vst[3][HD * 4];

	const int lane = lane_id(), wave = wave_id();
	const int j = lane & 31, hh = lane >> 5;
	const int kvh = blockIdx.x;
	const int hr = wave % HG, tw = wave / HG;
	const int b0 = (blockIdx.y * TW + tw) * 32;           // this wave's first token
	const int bw = blockIdx.y * TW * 32;                  // the workgroup's first token
	const int nb = a.pf_nb;
	const int kv_all = a.pf_kv0 + min(bw + TW * 32, nb);  // rows the workgroup's last token attends to
	const int ntiles = (kv_all + 31) >> 5;
	const int my_tiles = b0 < nb ? (a.pf_kv0 + min(b0 + 32, nb) + 31) >> 5 : 0;
	const float inv_sqrt_hd = 1.0f / sqrtf((float)a.head_dim);
	const int qpos = a.pf_kv0 + b0 + j; // this lane's query may look at rows <= qpos

	constexpr int EB = KVB / 8;
	const unsigned char* kbase = (const unsigned char*)a.kc + (size_t)kvh * a.seq_len * HD * EB;
	const unsigned char* vbase = (const unsigned char*)a.vc + (size_t)kvh * a.seq_len * HD * EB;
	// staging: a wave-load covers 64 / CK rows of 16-byte chunks (fp8: of 8-byte chunks); the workgroup's 256 lanes cover
	// 1024 / CK rows per pass, 32 rows in CK / 8 ... passes
	constexpr int RPP = 256 / CK;  // rows per pass of the whole workgroup
	constexpr int NP = 32 / RPP;   // passes per tile (HD 128: 2, HD 64: 1)
	const int srow = threadIdx.x / CK, sck = threadIdx.x % CK;
	u32x4 kreg[NP], vreg[NP];
	const int vd = threadIdx.x >> 2, vpc = threadIdx.x & 3; // VT: this lane's head dimension within a pass of 64, its quarter of the tile's 32 keys
	auto fetch = [&](int kt) {
#pragma unroll
		for (int p = 0; p < NP; ++p) {
			const int row = min(kt * 32 + p * RPP + srow, kv_all - 1);
			if constexpr (KVB == 16) {
				kreg[p] = *(const u32x4*)(kbase + ((size_t)row * HD + sck * 8) * 2);
				if constexpr (VT) { // a tile of 32 keys is one block of the transposed cache: [dim][32 keys], 64 bytes per dim
					vreg[p] = *(const u32x4*)(vbase + ((size_t)kt * HD + p * 64 + vd) * VT_BLOCK_BYTES + vpc * 16);
				} else {
					vreg[p] = *(const u32x4*)(vbase + ((size_t)row * HD + sck * 8) * 2);
				}
			} else { // e5m2 -> binary16: the byte becomes the upper byte
				const u32x2 kw = *(const u32x2*)(kbase + (size_t)row * HD + sck * 8);
				// VT: half a block of 64 keys, 32 bytes per dim, 8 keys per lane
				const u32x2 vw = VT ? *(const u32x2*)(vbase + ((size_t)(kt >> 1) * HD + p * 64 + vd) * VT_BLOCK_BYTES + (kt & 1) * 32 + vpc * 8)
				                    : *(const u32x2*)(vbase + (size_t)row * HD + sck * 8);
				kreg[p] = (u32x4){__builtin_amdgcn_perm(kw[0], kw[0], 0x050c040cu), __builtin_amdgcn_perm(kw[0], kw[0], 0x070c060cu),
				                  __builtin_amdgcn_perm(kw[1], kw[1], 0x050c040cu), __builtin_amdgcn_perm(kw[1], kw[1], 0x070c060cu)};
				vreg[p] = (u32x4){__builtin_amdgcn_perm(vw[0], vw[0], 0x050c040cu), __builtin_amdgcn_perm(vw[0], vw[0], 0x070c060cu),
				                  __builtin_amdgcn_perm(vw[1], vw[1], 0x050c040cu), __builtin_amdgcn_perm(vw[1], vw[1], 0x070c060cu)};
			}
		}
	};
	auto kswz = [](int key) { return HD == 128 ? (key & 15) : ((key >> 1) & 7); };
	auto stage = [&](int slot, int kt) {
#pragma unroll
		for (int p = 0; p < NP; ++p) {
			const int key = p * RPP + srow; // within the tile
			kst[slot][key * CK + (sck ^ kswz(key))] = kreg[p];
			if constexpr (VT) {
				// 8 consecutive keys (8 vpc .. + 7 of the tile) of head dimension d, as cached; keys past the rows anybody here attends to
				// carry P = 0 but may hold anything (a slot of an earlier, longer sequence; 0 x inf = NaN): cleared in the last tile
				const int d = p * 64 + vd;
				u32x4 v16 = vreg[p];
				const int live = kv_all - (kt * 32 + 8 * vpc);
				if (live < 8) {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						v16[i] = 2 * i + 1 < live ? v16[i] : (2 * i < live ? (v16[i] & 0xffffu) : 0u);
					}
				}
				vst[slot][d * 4 + (vpc ^ ((d >> 3) & 3))] = v16;
				continue;
			}
			// V transposed: this lane holds d = 8 sck .. + 7 of `key`; position of the key in the MFMA's order, 8-key chunk swizzled
			const int pi = (key & 16) + 8 * ((key >> 2) & 1) + (key & 3) + 4 * ((key >> 3) & 1);
			unsigned short* vt = (unsigned short*)vst[slot];
#pragma unroll
			for (int e = 0; e < 8; ++e) {
				const int d = 8 * sck + e;
				const unsigned w = vreg[p][e >> 1];
				vt[d * 32 + ((((pi >> 3) ^ ((d >> 3) & 3)) << 3) | (pi & 7))] = (unsigned short)((e & 1) ? (w >> 16) : (w & 0xffff));
			}
		}
	};

	fetch(0);
	stage(0, 0);
	if (ntiles > 1) {
		fetch(1);
	}
	__syncthreads();
	const int jk = VT ? ((j & ~12) | ((j & 8) >> 1) | ((j & 4) << 1)) : j; // the key (within a tile) this lane's A row of S^T holds

	for (int r = 0; r < a.kv_mul / HG; ++r) {
		const int h = kvh * a.kv_mul + r * HG + hr;
		// q of this lane's token: hi / lo operands, d = 16 t + 8 hh + e
		u32x4 qh[NT], ql[NT];
		{
			const float* qsrc = a.q + (size_t)min(b0 + j, nb - 1) * a.pf_stride + h * HD + 8 * hh;
#pragma unroll
			for (int t = 0; t < NT; ++t) {
				const float4 q0 = *(const float4*)(qsrc + 16 * t), q1 = *(const float4*)(qsrc + 16 * t + 4);
				unsigned hi, lo;
				pf_split2(q0.x, q0.y, hi, lo), qh[t][0] = hi, ql[t][0] = lo;
				pf_split2(q0.z, q0.w, hi, lo), qh[t][1] = hi, ql[t][1] = lo;
				pf_split2(q1.x, q1.y, hi, lo), qh[t][2] = hi, ql[t][2] = lo;
				pf_split2(q1.z, q1.w, hi, lo), qh[t][3] = hi, ql[t][3] = lo;
			}
		}
		f32x16 o[ND];
#pragma unroll
		for (int dt = 0; dt < ND; ++dt) {
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				o[dt][i] = 0.f;
			}
		}
		float m = -INFINITY, l = 0.f;

		for (int kt = 0; kt < ntiles; ++kt) {
			const int slot = kt % 3;
			// stage tile kt + 1 (fetched an iteration ago), fetch tile kt + 2
			if (kt + 1 < ntiles) {
				stage((kt + 1) % 3, kt + 1);
			}
			if (kt + 2 < ntiles) {
				fetch(kt + 2);
			}
			if (kt < my_tiles) {
				f32x16 sacc;
#pragma unroll
				for (int i = 0; i < 16; ++i) {
					sacc[i] = 0.f;
				}
#pragma unroll
				for (int t = 0; t < NT; ++t) {
					const f16x8 kop = __builtin_bit_cast(f16x8, kst[slot][jk * CK + ((2 * t + hh) ^ kswz(jk))]);
					sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kop, __builtin_bit_cast(f16x8, qh[t]), sacc, 0, 0, 0);
					sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kop, __builtin_bit_cast(f16x8, ql[t]), sacc, 0, 0, 0);
				}
				// scores of this lane's query against keys kt * 32 + (i & 3) + 8 (i >> 2) + 4 hh   (src/infer.c:244-248)
				// (VT: C row r = that expression holds key r with bits 2 and 3 exchanged: (i & 3) + 4 ((i >> 2) & 1) + 16 (i >> 3) + 8 hh)
				float mt = -INFINITY;
#pragma unroll
				for (int i = 0; i < 16; ++i) {
					const int key = kt * 32 + (VT ? (i & 3) + 4 * ((i >> 2) & 1) + 16 * (i >> 3) + 8 * hh : (i & 3) + 8 * (i >> 2) + 4 * hh);
					sacc[i] = key <= qpos ? sacc[i] * inv_sqrt_hd : -INFINITY;
					mt = fmaxf(mt, sacc[i]);
				}
				mt = fmaxf(mt, __shfl_xor(mt, 32));
				const float mn = fmaxf(m, mt); // finite from the first tile on: key 0 is visible to every query
				const float c = __expf(m - mn);
				float ls = 0.f;
#pragma unroll
				for (int i = 0; i < 16; ++i) {
					sacc[i] = __expf(sacc[i] - mn); // masked: exp(-inf) = 0
					ls += sacc[i];
				}
				ls += __shfl_xor(ls, 32);
				l = l * c + ls;
				m = mn;
				if (__any(c != 1.0f)) {
#pragma unroll
					for (int dt = 0; dt < ND; ++dt) {
#pragma unroll
						for (int i = 0; i < 16; ++i) {
							o[dt][i] *= c;
						}
					}
				}
				u32x4 ph[2], pl[2];
#pragma unroll
				for (int u = 0; u < 2; ++u) {
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const float p0 = sacc[8 * u + 2 * e], p1 = sacc[8 * u + 2 * e + 1];
						const __half2 hv = __floats2half2_rn(p0, p1); // 0 <= p <= 1
						const float2 hf = __half22float2(hv);
						ph[u][e] = __builtin_bit_cast(unsigned, hv);
						pl[u][e] = __builtin_bit_cast(unsigned, __floats2half2_rn(p0 - hf.x, p1 - hf.y));
					}
				}
#pragma unroll
				for (int dt = 0; dt < ND; ++dt) {
					const int d = 32 * dt + j;
#pragma unroll
					for (int u = 0; u < 2; ++u) {
						const f16x8 vop = __builtin_bit_cast(f16x8, vst[slot][d * 4 + ((2 * u + hh) ^ ((d >> 3) & 3))]);
						o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vop, __builtin_bit_cast(f16x8, ph[u]), o[dt], 0, 0, 0);
						o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vop, __builtin_bit_cast(f16x8, pl[u]), o[dt], 0, 0, 0);
					}
				}
			}
			__syncthreads();
		}
		// O^T tile dt: column = this lane's query, rows d = 32 dt + (i & 3) + 8 (i >> 2) + 4 hh   (src/infer.c:258-266)
		if (b0 + j < nb) {
			const int ns = pf_steps(a.pf_stride);
#pragma unroll
			for (int dt = 0; dt < ND; ++dt) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					const float v4[4] = {o[dt][4 * g] / l, o[dt][4 * g + 1] / l, o[dt][4 * g + 2] / l, o[dt][4 * g + 3] / l};
					pf_store4(a.out, b0 + j, h * HD + 32 * dt + 8 * g + 4 * hh, ns, v4);
				}
			}
		}
		// a further round of heads walks the same tiles again: restart the ring
		if (r + 1 < a.kv_mul / HG) {
			fetch(0);
			stage(0, 0);
			if (ntiles > 1) {
				fetch(1);
			}
			__syncthreads();
		}
	}
}

// log of the softmax probability of `target[b]` under row b of the logits (src/sampler.c:19-32 followed by the
// log of src/run.c:298): one workgroup per token.  target < 0: nothing to score, the slot gets 0.
__global__ __launch_bounds__(256) void k_pf_logprob(const float* logits, int vocab, const int* target, float* out) {
	__shared__ float red[16];
	const int b = blockIdx.x;
	const float* l = logits + (size_t)b * vocab;
	float mx = -3.402823466e+38f;
	for (int i = threadIdx.x; i < vocab; i += 256) {
		mx = fmaxf(mx, l[i]);
	}
	mx = wave_max(mx);
	__syncthreads();
	if (lane_id() == 0) {
		red[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
	float s = 0.f;
	for (int i = threadIdx.x; i < vocab; i += 256) {
		s += expf(l[i] - mx);
	}
	s = block_sum<256>(s, red);
	if (threadIdx.x == 0) {
		const int t = target[b];
		out[b] = t >= 0 ? (l[t] - mx) - logf(s) : 0.f;
	}
}

struct PfGemmArgs {
	const float4* xin;   // fragment-major activations (pf_unit), rows of K values as hi + lo binary16
	const void *w0, *w1, *w2; // QKV: wq, wk, wv;  FFN_UP: w1, w3;  RESID: the matrix
	int K, M, nb;        // reduction length, output units, valid tokens
	float* out;          // QKV: Q [token][q_dim];  RESID: X [token][M] (accumulated into);  FFN_UP: H, fragment-major rows of M
	const float* bqkv;
	const float2* rope;  // [token][head_dim / 2]
	void *kc, *vc;       // this layer's caches, [kv_head][seq_len][head_dim]
	void* vt;            // the transposed V cache [kv_head][head_dim][seq_len] (kernels.hip.h k_attn_vt), or nullptr
	int q_dim, kv_dim, head_dim, seq_len, kv_pos0;
	float clip;
	int gelu;
	// mixture of experts: a grouped GEMM over the packed rows of all experts (k_pf_route); workgroup column c
	// multiplies by the matrices of expert col_expert[c] (w0 / w1 + expert * expert_stride bytes)
	const int* col_expert;
	size_t expert_stride;
	int ncols; // k_pf_gemm_wide: token columns of the launch (its grid is one-dimensional)
	// k_pf_gemm_wide with too few tiles to fill the chip: K is cut into ksplit ranges, one workgroup each; partial tiles go to
	// `partial` ([tile][range][wave][register][lane] floats), the last of a tile's workgroups to bump tile_count[tile] adds them
	// in range order (deterministic) and runs the epilogue
	int ksplit;
	float* partial;
	unsigned* tile_count;
};

// operand j of a 16-byte piece of a weight row: 8 consecutive weights as binary16 (exact in all three formats).
// A piece holds G / 8 operands: one (fp16), two (fp8), four (gf4: one per word).
template <int DB>
__device__ __forceinline__ f16x8 pf_operand(u32x4 v, int j) {
	u32x4 r;
	if constexpr (DB == 16) {
		r = v;
	} else if constexpr (DB == 8) { // e5m2 is the upper byte of binary16 (src/infer.c:28-31)
		const unsigned w0 = v[2 * j], w1 = v[2 * j + 1];
		r[0] = __builtin_amdgcn_perm(w0, w0, 0x050c040cu);
		r[1] = __builtin_amdgcn_perm(w0, w0, 0x070c060cu);
		r[2] = __builtin_amdgcn_perm(w1, w1, 0x050c040cu);
		r[3] = __builtin_amdgcn_perm(w1, w1, 0x070c060cu);
	} else { // src/infer.c:33-40: w_k = (code_k - 4) * s, s = fp8(low byte) / -4; both factors and the product are exact in binary16
		const unsigned w = v[j];
		const __half sh = __float2half_rn(bf8_byte0(w) * -0.25f);
		const __half2 s2 = __halves2half2(sh, sh);
		const __half2 off = __builtin_bit_cast(__half2, 0xe404e404u); // (-1028, -1028)
#pragma unroll
		for (int pr = 0; pr < 4; ++pr) {
			// codes 2pr, 2pr+1 -> the low mantissa bits of (1024, 1024): 1024 + code, an integer
			const unsigned two = (w >> (8 + 6 * pr)) & 0x3fu;
			const unsigned h2 = ((two * 0x2001u) & 0x00070007u) | 0x64006400u;
			r[pr] = __builtin_bit_cast(unsigned, __hmul2(__hadd2(__builtin_bit_cast(__half2, h2), off), s2));
		}
	}
	return __builtin_bit_cast(f16x8, r);
}

// What happens to four consecutive output units ub .. ub + 3 (ub a multiple of 4, ub < M) of one token -- everything but the
// FFN-up's gated activation, which needs two accumulators.
template <int KVB, int EPI>
__device__ __forceinline__ void pf_epi4(const PfGemmArgs& a, const int token, const int ub, const float (&v)[4]) {
	if constexpr (EPI == PF_EPI_RESID) {
		float4* p = (float4*)(a.out + (size_t)token * a.M + ub);
		float4 t = *p;
		t.x += v[0], t.y += v[1], t.z += v[2], t.w += v[3];
		*p = t;
	} else if constexpr (EPI == PF_EPI_STORE) { // M is arbitrary here (a vocabulary): bounded; one 16-byte store where the rows allow it
		if ((a.M & 3) == 0) {
			*(float4*)(a.out + (size_t)token * a.M + ub) = make_float4(v[0], v[1], v[2], v[3]);
		} else {
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				if (ub + e < a.M) {
					a.out[(size_t)token * a.M + ub + e] = v[e];
				}
			}
		}
	} else {
		static_assert(EPI == PF_EPI_QKV, "the FFN-up epilogue is not per accumulator");
		float r[4];
#pragma unroll
		for (int pr = 0; pr < 2; ++pr) { // RoPE pairs (2i, 2i+1); q / k / v boundaries are multiples of 8
			const int uu = ub + 2 * pr;
			float v0 = v[2 * pr], v1 = v[2 * pr + 1];
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
			r[2 * pr] = v0, r[2 * pr + 1] = v1;
		}
		if (ub < a.q_dim) {
			*(float4*)(a.out + (size_t)token * a.q_dim + ub) = make_float4(r[0], r[1], r[2], r[3]);
		} else {
			int jl = ub - a.q_dim;
			void* cache = a.kc;
			if (jl >= a.kv_dim) {
				jl -= a.kv_dim;
				cache = a.vc;
			}
			const size_t off = ((size_t)(jl / a.head_dim) * a.seq_len + a.kv_pos0 + token) * a.head_dim + (jl % a.head_dim);
			const bool tr = cache == a.vc && a.vt;
			const size_t offt = tr ? attn_vt_offset(jl, a.kv_pos0 + token, a.head_dim, a.seq_len, KVB / 8) : 0; // the same elements in the transposed cache: dims jl .. jl + 3
			constexpr int nd = VT_BLOCK_BYTES / (KVB / 8);                                                             // (the next dim of the same position)
			if constexpr (KVB == 16) { // src/infer.c:378-381
				const __half2 lo = __floats2half2_rn(r[0], r[1]), hi = __floats2half2_rn(r[2], r[3]);
				*(u32x2*)((__half*)cache + off) = (u32x2){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
				if (tr) {
					__half* t = (__half*)a.vt + offt;
					t[0] = __low2half(lo), t[nd] = __high2half(lo), t[2 * nd] = __low2half(hi), t[3 * nd] = __high2half(hi);
				}
			} else {
				const unsigned w = (unsigned)e5m2x2_sat(r[0], r[1]) | ((unsigned)e5m2x2_sat(r[2], r[3]) << 16);
				*(unsigned*)((unsigned char*)cache + off) = w;
				if (tr) {
					unsigned char* t = (unsigned char*)a.vt + offt;
					t[0] = (unsigned char)w, t[nd] = (unsigned char)(w >> 8), t[2 * nd] = (unsigned char)(w >> 16), t[3 * nd] = (unsigned char)(w >> 24);
				}
			}
		}
	}
}

// Epilogue of a wave's NA x NC accumulator tiles (32 units x 32 tokens each; FFN-up: NA = 2 are w1 and w3 of one strip),
// straight from the accumulator registers: a lane stores for ITS token (one cache line per lane and store).
// C layout: column (token) = lane & 31, row (unit) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <int KVB, int EPI, int NA, int NC>
__device__ __forceinline__ void pf_epilogue(const PfGemmArgs& a, const f32x16 (&acc)[NA][NC], const int unit0, const int tok0, const int j, const int kk) {
	const int nb = a.nb;
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int token = tok0 + 32 * c + j;
		if (token >= nb) {
			continue;
		}
#pragma unroll
		for (int n = 0; n < (EPI == PF_EPI_FFN_UP ? 1 : NA); ++n) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const int ub = unit0 + 32 * n + 8 * g + 4 * kk; // four consecutive units; M is a multiple of 4 (a vocabulary may not be)
				if (ub >= a.M) {
					continue;
				}
				if constexpr (EPI == PF_EPI_FFN_UP) {
					float h[4];
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						float up = acc[0][c][4 * g + e], gt = acc[1][c][4 * g + e];
						h[e] = (a.gelu ? act_gelu(up) : act_silu(up)) * gt; // src/infer.c:440-450
					}
					pf_store4(a.out, token, ub, pf_steps(a.M), h);
				} else {
					const float v[4] = {acc[n][c][4 * g], acc[n][c][4 * g + 1], acc[n][c][4 * g + 2], acc[n][c][4 * g + 3]};
					pf_epi4<KVB, EPI>(a, token, ub, v);
				}
			}
		}
	}
}

// The same through a wave-private LDS image [32 tokens][32 NA units (+ 4)]: a store instruction then covers whole runs of
// consecutive units of a few tokens -- eight cache lines instead of sixty-four (the token-per-lane stores were 8-20 % of a GEMM,
// profiles/r02_prefill_gemm.txt); the RoPE table reads and the KV-cache stores of the QKV epilogue coalesce the same way.
// `img`: 32 x (32 NA + 4) floats of LDS nobody else uses.  Not for the FFN-up (its fragment-major stores are contiguous as they are).
template <int KVB, int EPI, int NA>
__device__ __forceinline__ void pf_epilogue_rows(const PfGemmArgs& a, const f32x16 (&acc)[NA][2], const int unit0, const int tok0, float* img) {
	constexpr int RS = 32 * NA + 4;     // row stride of the image (floats)
	constexpr int CH = 8 * NA;          // float4 chunks per token row
	constexpr int RPI = 64 / CH;        // token rows per pass of the wave (NA = 3: two rows, 48 lanes)
	const int lane = lane_id(), j = lane & 31, kk = lane >> 5;
#pragma unroll
	for (int c = 0; c < 2; ++c) {
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				*(float4*)(img + j * RS + 32 * n + 8 * g + 4 * kk) = make_float4(acc[n][c][4 * g], acc[n][c][4 * g + 1], acc[n][c][4 * g + 2], acc[n][c][4 * g + 3]);
			}
		}
		const int ub = unit0 + 4 * (lane % CH);
#pragma unroll
		for (int it = 0; it < 32 / RPI; ++it) {
			const int row = RPI * it + lane / CH, token = tok0 + 32 * c + row;
			if (lane < RPI * CH) {
				const float4 t = *(const float4*)(img + row * RS + 4 * (lane % CH));
				if (token < a.nb && ub < a.M) {
					const float v[4] = {t.x, t.y, t.z, t.w};
					pf_epi4<KVB, EPI>(a, token, ub, v);
				}
			}
		}
	}
}

// ---- chunks of 3 or 4 tokens: every weight streamed ONCE, at the decode kernels' rate ----------------------------------------------
// The GEMM forms below cost a flat ~160 us per layer for a handful of tokens (every weight streamed once at 1.4 TB/s), four serial
// decode steps 200.  k_pf_skinny is the decode row engine's stream with T tokens' activations behind it
// (tools/experiments/exp_skinny.hip, profiles/r03_multi_token_probe.txt: four tokens in 1.55-2.3 x one token's launch):
//   * a wave-load covers 256 bytes of each of FOUR weight rows (lane 4 b + n: bytes [16 b, 16 b + 16) of row n's chunk) -- the operand
//     layout of v_mfma_f32_4x4x4_16b_f16: sixteen independent 4 x 4 x 4 products per wave, block b = lane / 4, B column n = lane % 4 =
//     the weight ROW, A row i = lane % 4 = the TOKEN, K = 4 weights per instruction;
//   * the weights as binary16 (pf_operand: exact in all three formats), the activations from the same fragment-major hi + lo
//     matrices the GEMMs read (pf_unit), re-laid in LDS per (chunk, K-step, block, token): 16 bytes = 4 hi | 4 lo halves; two MFMAs
//     per K-step, fp32 accumulation in the lane's four D registers (one per token) across the whole row;
//   * one cross-lane sum per group of four rows, then the GEMMs' own epilogues (pf_epi4 / the gated activation + pf_store4) per token.
// T = 4 tokens in the product (tokens >= nb are zero and dropped; the T = 8 instantiation measured slower than the GEMM forms and is
// not launched); [k0, k0 + kn) = the columns this launch covers -- whole chunks; a residual
// GEMM whose image does not fit the LDS runs as several launches over column ranges, each adding its part (pf_epi4 accumulates).
// grid: like the decode kernels', groups of four units dealt round-robin over the waves; LDS = T * kn * 4 bytes.
template <int DB, int KVB, int EPI, int T>
__global__ __launch_bounds__(256) void k_pf_skinny(PfGemmArgs a, int k0, int kn) {
	static_assert(EPI == PF_EPI_QKV || EPI == PF_EPI_RESID || EPI == PF_EPI_FFN_UP, "dense GEMMs of a layer");
	constexpr int G = Fmt<DB>::G;   // weights per 16-byte lane-load
	constexpr int SPL = G / 4;      // MFMA K-steps (4 weights) per lane-load
	constexpr int CC = 16 * G;      // columns per chunk: one wave-load of each of the four rows
	constexpr int NA = EPI == PF_EPI_FFN_UP ? 2 : 1; // weight streams: FFN-up multiplies w1 and w3 with the same activations
	constexpr int TS = T / 4;       // token sets of four
	constexpr int U = 4 / NA;       // chunks per tile: four wave-loads, two tiles in flight per wave
	extern __shared__ __attribute__((aligned(16))) unsigned char pf_sk_smem[];
	u32x4* img = (u32x4*)pf_sk_smem;
	const int lane = lane_id(), wave = wave_id();
	const int b = lane >> 2, n = lane & 3;
	const int nchunks = kn / CC;
	const int ngroups = a.M / 4;
	const int stride = gridDim.x * 4;
	const size_t row_bytes = (size_t)a.K * DB / 8;
	const size_t col_off = (size_t)k0 * DB / 8;

	// row n of group g in stream s
	auto row_of = [&](int g, int s) -> const unsigned char* {
		const int u = min(g, ngroups - 1) * 4 + n;
		if constexpr (EPI == PF_EPI_QKV) {
			const bool is_q = u < a.q_dim, is_k = u < a.q_dim + a.kv_dim;
			const unsigned char* base = (const unsigned char*)(is_q ? a.w0 : (is_k ? a.w1 : a.w2));
			return base + (size_t)(u - (is_q ? 0 : (is_k ? a.q_dim : a.q_dim + a.kv_dim))) * row_bytes + col_off;
		} else {
			return (const unsigned char*)(s == 0 ? a.w0 : a.w1) + (size_t)u * row_bytes + col_off;
		}
	};
	u32x4 tile[2][NA][U];
	auto load = [&](int ph, int g, int c0) {
#pragma unroll
		for (int s = 0; s < NA; ++s) {
			const unsigned char* row = row_of(g, s);
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int c = min(c0 + u, nchunks - 1);
				tile[ph][s][u] = __builtin_nontemporal_load((gptr16)(row + (size_t)c * 256) + b);
			}
		}
	};
	int grp = blockIdx.x * 4 + wave, c0 = 0;
	int grp1 = grp, c1 = U;
	if (c1 >= nchunks) {
		c1 = 0, grp1 += stride;
	}
	load(0, grp, c0);
	load(1, grp1, c1);

	// the image: per (token t, 8 columns) one hi and one lo unit of the fragment-major matrix -> two slots (K-steps 2 j, 2 j + 1)
	{
		const int nsteps = pf_steps(a.K);
		const u32x4* xin = (const u32x4*)a.xin;
		for (int idx = threadIdx.x; idx < T * (kn / 8); idx += 256) {
			const int t = idx / (kn / 8), kk = (idx % (kn / 8)) * 8;
			u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
			if (t < a.nb) {
				const int un = pf_unit(t, k0 + kk, nsteps);
				hi = xin[un], lo = xin[un + 64];
			}
			const int c = kk / CC, col = kk % CC, bb = col / G, s0 = (col % G) / 4;
			u32x4* slot = img + ((c * SPL + s0) * 16 + bb) * T + t;
			slot[0] = (u32x4){hi[0], hi[1], lo[0], lo[1]};
			slot[16 * T] = (u32x4){hi[2], hi[3], lo[2], lo[3]};
		}
	}
	__syncthreads();

	f32x4 acc[NA][TS];
#pragma unroll
	for (int s = 0; s < NA; ++s) {
#pragma unroll
		for (int q = 0; q < TS; ++q) {
			acc[s][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
		}
	}
	while (grp < ngroups) {
#pragma unroll
		for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (c0 + u < nchunks) { // wave-uniform
#pragma unroll
					for (int j = 0; j < G / 8; ++j) {
						f16x8 wop[NA];
#pragma unroll
						for (int s = 0; s < NA; ++s) {
							wop[s] = pf_operand<DB>(tile[ph][s][u], j);
						}
#pragma unroll
						for (int h = 0; h < 2; ++h) { // K-step 2 j + h: weights 4 h .. 4 h + 3 of the operand
#pragma unroll
							for (int q = 0; q < TS; ++q) {
								const u32x4 av = img[(((c0 + u) * SPL + 2 * j + h) * 16 + b) * T + 4 * q + n]; // this lane's A row = token 4 q + n
								const u32x2 ah = {av[0], av[1]}, al = {av[2], av[3]};
#pragma unroll
								for (int s = 0; s < NA; ++s) {
									const u32x4 w4 = __builtin_bit_cast(u32x4, wop[s]);
									const u32x2 bw = {w4[2 * h], w4[2 * h + 1]};
									acc[s][q] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, ah), __builtin_bit_cast(f16x4, bw), acc[s][q], 0, 0, 0);
									acc[s][q] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, al), __builtin_bit_cast(f16x4, bw), acc[s][q], 0, 0, 0);
								}
							}
						}
					}
				}
			}
			const bool last = c0 + U >= nchunks;
			const int g_done = grp;
			int g2 = grp1, c2 = c1 + U;
			if (c2 >= nchunks) {
				c2 = 0, g2 += stride;
			}
			load(ph, g2, c2);
			if (last) {
				// D register i of lane 4 b + n = token 4 q + i, row n, summed over block b's columns: add the sixteen blocks (lanes 60 + n hold the sums)
				float sum[NA][T];
#pragma unroll
				for (int s = 0; s < NA; ++s) {
#pragma unroll
					for (int q = 0; q < TS; ++q) {
#pragma unroll
						for (int i = 0; i < 4; ++i) {
							float v = acc[s][q][i];
							v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)); // row_shr:4
							v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true)); // row_shr:8
							v += __shfl_xor(v, 16);
							v += __shfl_xor(v, 32);
							sum[s][4 * q + i] = v;
						}
						acc[s][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
					}
				}
				const int ub = g_done * 4;
#pragma unroll
				for (int t = 0; t < T; ++t) {
					if (t < a.nb) { // uniform
						float r[NA][4];
#pragma unroll
						for (int s = 0; s < NA; ++s) {
#pragma unroll
							for (int e = 0; e < 4; ++e) {
								r[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sum[s][t]), 60 + e));
							}
						}
						if (lane == 0) {
							if constexpr (EPI == PF_EPI_FFN_UP) {
								float hv[4];
#pragma unroll
								for (int e = 0; e < 4; ++e) {
									hv[e] = (a.gelu ? act_gelu(r[0][e]) : act_silu(r[0][e])) * r[NA - 1][e]; // src/infer.c:440-450
								}
								pf_store4(a.out, t, ub, pf_steps(a.M), hv);
							} else {
								const float v4[4] = {r[0][0], r[0][1], r[0][2], r[0][3]};
								pf_epi4<KVB, EPI>(a, t, ub, v4);
							}
						}
					}
				}
			}
			grp = grp1, c0 = c1;
			grp1 = g2, c1 = c2;
			if (grp >= ngroups) {
				break;
			}
		}
	}
}

// Weight streams per wave: S strips of 32 units (S = 1..3, whichever wastes the fewest workgroup rounds), except
// FFN-up whose two streams are w1 and w3 of ONE strip.
template <int EPI, int S>
struct PfTile {
	static constexpr int NA = EPI == PF_EPI_FFN_UP ? 2 : S;
	static constexpr int UNITS = EPI == PF_EPI_FFN_UP ? 32 : 32 * S; // output units per workgroup
};

// grid = (ceil(M / UNITS), ceil(nb / 64)), 256 threads
template <int DB, int KVB, int EPI, int S>
__global__ __launch_bounds__(256, S < 3 ? 2 : 1) void k_pf_gemm(PfGemmArgs a) {
	constexpr int G = Fmt<DB>::G;
	constexpr int P = 32 / G; // 16-byte pieces of a row per lane and 64-column step (32 weights per k-half)
	constexpr int OPP = G / 8; // MFMA operands (8 weights) per piece
	// a wave's unit of work is a SUB-step: 1 / HS of a step (16 weights per k-half for fp16 / fp8), which halves
	// the operand registers in flight so that two workgroups fit a CU (one wave per SIMD left every stall exposed)
	constexpr int HS = (P >= 2 && S < 3) ? 2 : 1, PH = P / HS, QH = 8 / HS; // QH = 2 (hi, lo) * PH * OPP units of B per token tile
	constexpr int NA = PfTile<EPI, S>::NA, NC = 2;
	__shared__ float part[2][NA * NC * 16][64]; // partial tiles in flight during the two-round reduction (16 KiB per stream)

	const int lane = lane_id(), wave = wave_id();
	const int j = lane & 31, kk = lane >> 5;
	const int unit0 = blockIdx.x * PfTile<EPI, S>::UNITS, tok0 = blockIdx.y * 64;
	size_t expert_off = 0;
	if (a.col_expert) {
		const int e = a.col_expert[blockIdx.y];
		if (e < 0) {
			return; // the grid is sized for the worst-case number of columns
		}
		expert_off = (size_t)e * a.expert_stride;
	}
	const size_t row_bytes = (size_t)a.K * DB / 8;
	const int npieces = a.K / G;      // 16-byte pieces per row
	const int nsteps = pf_steps(a.K); // a step = 64 weights of a row = 4 (x hi, lo) MFMAs per accumulator tile

	// A: this lane's weight rows.  Row indices are clamped, never branched on: surplus lanes read real data
	// and their results are dropped in the epilogue.
	const unsigned char* rowp[NA];
#pragma unroll
	for (int s = 0; s < NA; ++s) {
		if constexpr (EPI == PF_EPI_QKV) {
			const int u = min(unit0 + 32 * s + j, a.M - 1);
			const bool is_q = u < a.q_dim, is_k = u < a.q_dim + a.kv_dim;
			const unsigned char* base = (const unsigned char*)(is_q ? a.w0 : (is_k ? a.w1 : a.w2));
			const int ul = u - (is_q ? 0 : (is_k ? a.q_dim : a.q_dim + a.kv_dim));
			rowp[s] = base + (size_t)ul * row_bytes;
		} else if constexpr (EPI == PF_EPI_FFN_UP) {
			rowp[s] = (const unsigned char*)(s ? a.w1 : a.w0) + expert_off + (size_t)min(unit0 + j, a.M - 1) * row_bytes;
		} else {
			rowp[s] = (const unsigned char*)a.w0 + expert_off + (size_t)min(unit0 + 32 * s + j, a.M - 1) * row_bytes;
		}
	}
	// B: the two 32-token groups of this workgroup (the matrix is allocated for whole groups of 64 tokens)
	const float4* xg = a.xin + (size_t)(tok0 >> 5) * nsteps * 512 + lane;

	struct Frag {
		u32x4 w[NA][PH];
		u32x4 x[NC][QH]; // [MFMA of the sub-step][hi, lo]
	};
	const int nsub = nsteps * HS;
	auto load = [&](Frag& f, int u) {
		const int uc = min(u, nsub - 1);
		const int sc = uc / HS, h = uc % HS;
		const int p0 = (2 * sc + kk) * P + h * PH;
#pragma unroll
		for (int i = 0; i < PH; ++i) {
			const int piece = min(p0 + i, npieces - 1);
#pragma unroll
			for (int n = 0; n < NA; ++n) {
				f.w[n][i] = __builtin_nontemporal_load((gptr16)rowp[n] + piece);
			}
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const u32x4* xp = (const u32x4*)(xg + ((size_t)c * nsteps + sc) * 512 + h * QH * 64);
#pragma unroll
			for (int q = 0; q < QH; ++q) {
				f.x[c][q] = xp[q * 64];
			}
		}
	};

	f32x16 acc[NA][NC];
#pragma unroll
	for (int n = 0; n < NA; ++n) {
#pragma unroll
		for (int c = 0; c < NC; ++c) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				acc[n][c][r] = 0.f;
			}
		}
	}
	auto compute = [&](const Frag& f, int u) {
		const int p0 = (2 * (u / HS) + kk) * P + (u % HS) * PH;
#pragma unroll
		for (int i = 0; i < PH; ++i) {
			const bool valid = p0 + i < npieces; // ragged rows: pieces past the row's end multiply as zeros
			u32x4 v[NA];
#pragma unroll
			for (int n = 0; n < NA; ++n) {
				v[n] = f.w[n][i];
				if (!valid) {
					v[n] = (u32x4){0u, 0u, 0u, 0u}; // decodes to zeros in every format
				}
			}
#pragma unroll
			for (int j = 0; j < OPP; ++j) {
				f16x8 wa[NA];
#pragma unroll
				for (int n = 0; n < NA; ++n) {
					wa[n] = pf_operand<DB>(v[n], j);
				}
#pragma unroll
				for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
					for (int n = 0; n < NA; ++n) {
#pragma unroll
						for (int c = 0; c < NC; ++c) {
							acc[n][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[n], __builtin_bit_cast(f16x8, f.x[c][(i * OPP + j) * 2 + hl]), acc[n][c], 0, 0, 0);
						}
					}
				}
			}
		}
	};

	// wave w takes sub-steps w, w+4, ...  The loop is unrolled by two so the two operand buffers alternate by name
	// (no register copies, which would wait for the loads just issued), and a scheduling barrier after each
	// load block keeps the compiler from sinking the loads down to their first use (it did: vmcnt(0) in front
	// of every other MFMA).  Loads are clamped, never skipped, so s_waitcnt stays counted.
	Frag f0, f1;
	load(f0, wave);
	for (int u = wave; u < nsub; u += 8) {
		load(f1, u + 4);
		__builtin_amdgcn_sched_barrier(0);
		compute(f0, u);
		load(f0, u + 8);
		__builtin_amdgcn_sched_barrier(0);
		if (u + 4 < nsub) {
			compute(f1, u + 4);
		}
	}

	// add the four waves' partial tiles, (w0 + w2) + (w1 + w3): a fixed order, two rounds through LDS
	auto put = [&](int slot) {
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					part[slot][(n * NC + c) * 16 + r][lane] = acc[n][c][r];
				}
			}
		}
	};
	auto add = [&](int slot) {
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					acc[n][c][r] += part[slot][(n * NC + c) * 16 + r][lane];
				}
			}
		}
	};
	if (wave >= 2) {
		put(wave - 2);
	}
	__syncthreads();
	if (wave >= 2) {
		return;
	}
	add(wave);
	__syncthreads(); // uniform for the two remaining waves: both slots have been read
	if (wave == 1) {
		put(0);
	}
	__syncthreads();
	if (wave == 1) {
		return;
	}
	add(0);

	if constexpr (EPI == PF_EPI_FFN_UP) {
		pf_epilogue<KVB, EPI, NA, NC>(a, acc, unit0, tok0, j, kk);
	} else {
		if (a.M & 3) { // a vocabulary that is not a multiple of 4: rows are not 16-byte aligned
			pf_epilogue<KVB, EPI, NA, NC>(a, acc, unit0, tok0, j, kk);
		} else { // part[1] is free: its last reader was wave 1, two barriers ago
			pf_epilogue_rows<KVB, EPI, NA>(a, acc, unit0, tok0, &part[1][0][0]);
		}
	}
}

// ---- the wide form ------------------------------------------------------------------------------------------------------
// k_pf_gemm splits K over the four waves of a workgroup to fill the chip from few tiles; each wave then fetches its own
// operands, 0.08 bytes per multiply-add, and at the f16 MFMA rate the kernel runs at whatever the CU's vector memory path
// keeps in flight (~64 KiB per CU: 8-10 TB/s of L2 traffic, 160-280 TFLOP/s).  When a GEMM has enough tiles to fill the chip
// WITHOUT splitting K, the four waves take different unit strips of the SAME 64 tokens and the same k: the B operand -- four
// fifths of the operand bytes -- is fetched once per workgroup, staged through LDS (a ring of three 16 KiB steps, already in
// the order the lanes consume it: the fragment-major matrix is copied verbatim) and read by all four waves with
// ds_read_b128; A goes global -> registers as before.  0.03 bytes per multiply-add, no cross-wave reduction.
// Workgroup tile: 256 units x 64 tokens (FFN-up: 128 units of w1 and of w3).
// Grid: one dimension, 8 * ceil(nx / 8) * ny workgroups for nx unit blocks and ny token columns.  Workgroups go to the 8 XCDs
// round robin; XCD c takes the unit blocks c, c + 8, ... and walks the token columns of one unit block before the next, so
// that the workgroups sharing a slice of weights run at the same time on the same L2 (the slice comes from HBM once) and
// move through k together; the token columns' B slices are shared the same way by the unit blocks in flight on the XCD.
// Short prompts (too few tiles): x ksplit workgroups per tile, each a range of K; nobody waits for anybody -- the last of a
// tile's workgroups to arrive folds the partial tiles in range order and runs the epilogue.
template <int EPI>
struct PfWide {
	static constexpr int UNITS = EPI == PF_EPI_FFN_UP ? 128 : 256;
};
__host__ __device__ inline int pf_wide_grid(int nx, int ny, int ksplit = 1) {
	return 8 * ((nx + 7) / 8) * ny * ksplit;
}

// A step of A (64 columns of this wave's 64 weight rows) is fetched row-contiguous -- a wave-load covers whole 32 / 64 / 128-byte
// row segments, 8-32 cache lines instead of one line per lane -- and turned into the MFMA's lane order through a wave-private
// LDS image (row stride padded by 16 bytes: the ds_read_b128 operand fetches are conflict free for the 16-lane groups).
// No barrier is involved: a wave's LDS operations execute in order.
template <int DB>
struct PfWideA {
	static constexpr int BR = 64 * DB / 8;       // bytes of a row per step: 32 (gf4), 64 (fp8), 128 (fp16)
	static constexpr int PR = BR / 16;           // 16-byte pieces per row = wave-loads per step
	static constexpr int RS = BR + 16;           // row stride of the image
	static constexpr int WAVE_BYTES = 64 * RS;
	static constexpr int LDS_BYTES = 3 * 16 * 1024 + 4 * WAVE_BYTES; // B ring + the four waves' A images
};

template <int DB, int KVB, int EPI, int AA>
__global__ __launch_bounds__(256, 2) void k_pf_gemm_wide(PfGemmArgs a) {
	constexpr int G = Fmt<DB>::G;
	constexpr int P = 32 / G;  // 16-byte pieces of a row per lane (k-half) and 64-column step
	constexpr int OPP = G / 8; // MFMA operands per piece
	constexpr int NA = 2, NC = 2;
	constexpr int PR = PfWideA<DB>::PR, RS = PfWideA<DB>::RS;
	constexpr int RPL = 64 / PR;  // rows per wave-load
	constexpr int AB = 2;         // B is fetched two steps ahead (it is staged one step before its use)
	constexpr int NWB = AA + 1;   // A is fetched AA steps ahead
	constexpr int U = AB * NWB / (NWB % 2 == 0 ? 2 : 1); // lcm(AB, NWB): the loop is unrolled so that every buffer has a static name
	extern __shared__ u32x4 pfw_lds[];
	u32x4(*bst)[16][64] = (u32x4(*)[16][64])pfw_lds; // [3][16][64]

	const int lane = lane_id(), wave = wave_id();
	const int j = lane & 31, kk = lane >> 5;
	unsigned char* const aimg = (unsigned char*)(pfw_lds + 3 * 16 * 64) + wave * PfWideA<DB>::WAVE_BYTES;
	const int ny = a.ncols, KS = a.ksplit;
	// order within an XCD: unit block, then K range, then token column
	const int idx = blockIdx.x >> 3;
	const int bx = (blockIdx.x & 7) + 8 * (idx / (ny * KS)), by = idx % ny, ks = (idx / ny) % KS;
	if (bx * PfWide<EPI>::UNITS >= a.M) {
		return;
	}
	const int unit0 = bx * PfWide<EPI>::UNITS + wave * (EPI == PF_EPI_FFN_UP ? 32 : 64), tok0 = by * 64;
	size_t expert_off = 0;
	if (a.col_expert) {
		const int e = a.col_expert[by];
		if (e < 0) {
			return;
		}
		expert_off = (size_t)e * a.expert_stride;
	}
	const size_t row_bytes = (size_t)a.K * DB / 8;
	const int row_pieces = (int)(row_bytes / 16);
	const int nsteps = pf_steps(a.K);
	const int s_begin = (int)((long)ks * nsteps / KS), s_end = (int)((long)(ks + 1) * nsteps / KS); // this workgroup's steps

	// wave-load q of a step: this lane fetches piece (lane % PR) of row q * RPL + lane / PR of the wave's 64 rows
	// (rows 0-31: first strip, 32-63: second strip -- FFN-up: w1 and w3 of one strip).  Clamped, never branched on.
	const int apiece = lane % PR;
	const unsigned char* rowq[PR];
#pragma unroll
	for (int q = 0; q < PR; ++q) {
		const int r = q * RPL + lane / PR;
		if constexpr (EPI == PF_EPI_QKV) {
			const int u = min(unit0 + r, a.M - 1);
			const bool is_q = u < a.q_dim, is_k = u < a.q_dim + a.kv_dim;
			const unsigned char* base = (const unsigned char*)(is_q ? a.w0 : (is_k ? a.w1 : a.w2));
			const int ul = u - (is_q ? 0 : (is_k ? a.q_dim : a.q_dim + a.kv_dim));
			rowq[q] = base + (size_t)ul * row_bytes;
		} else if constexpr (EPI == PF_EPI_FFN_UP) {
			rowq[q] = (const unsigned char*)(r >= 32 ? a.w1 : a.w0) + expert_off + (size_t)min(unit0 + (r & 31), a.M - 1) * row_bytes;
		} else {
			rowq[q] = (const unsigned char*)a.w0 + expert_off + (size_t)min(unit0 + r, a.M - 1) * row_bytes;
		}
	}
	const float4* xg = a.xin + (size_t)(tok0 >> 5) * nsteps * 512 + lane;

	u32x4 fb[AB][4];   // this wave's quarter of a step of B on its way to LDS: rows 4 * wave .. + 3 of [token group c][unit u]
	u32x4 fa[NWB][PR]; // a step of the wave's A rows on its way to the wave's LDS image
	auto load_b = [&](u32x4 (&b)[4], int sc) {
		const int scc = min(sc, nsteps - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int row = wave * 4 + r;
			b[r] = *(const u32x4*)(xg + ((size_t)(row >> 3) * nsteps + scc) * 512 + (row & 7) * 64);
		}
	};
	auto load_a = [&](u32x4 (&w)[PR], int sc) {
		const int piece = min(min(sc, nsteps - 1) * PR + apiece, row_pieces - 1);
#pragma unroll
		for (int q = 0; q < PR; ++q) {
			w[q] = __builtin_nontemporal_load((gptr16)rowq[q] + piece);
		}
	};
	auto stage_b = [&](const u32x4 (&b)[4], int slot) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			bst[slot][wave * 4 + r][lane] = b[r];
		}
	};
	auto stage_a = [&](const u32x4 (&w)[PR], int sc) {
		const bool valid = sc * PR + apiece < row_pieces; // ragged rows: pieces past the row's end multiply as zeros
#pragma unroll
		for (int q = 0; q < PR; ++q) {
			*(u32x4*)(aimg + (q * RPL + lane / PR) * RS + apiece * 16) = valid ? w[q] : (u32x4){0u, 0u, 0u, 0u};
		}
	};

	f32x16 acc[NA][NC];
#pragma unroll
	for (int n = 0; n < NA; ++n) {
#pragma unroll
		for (int c = 0; c < NC; ++c) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				acc[n][c][r] = 0.f;
			}
		}
	}
	// The B operands of MFMA group m + 1 (four ds_read_b128: two token tiles x hi, lo) are asked for before the eight MFMAs of
	// group m are issued -- 256 matrix-core cycles cover the LDS round trip; left to itself the compiler issues each read two
	// MFMAs ahead of its use and every group starts with a stall.
	auto compute = [&](int slot) {
		u32x4 w[NA][P];
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int i = 0; i < P; ++i) {
				w[n][i] = *(const u32x4*)(aimg + (32 * n + j) * RS + (kk * P + i) * 16);
			}
		}
		u32x4 bq[2][4];
		auto read_b = [&](u32x4 (&q)[4], int m) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int hl = 0; hl < 2; ++hl) {
					q[c * 2 + hl] = bst[slot][c * 8 + m * 2 + hl][lane];
				}
			}
		};
		read_b(bq[0], 0);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			if (m + 1 < 4) {
				read_b(bq[(m + 1) & 1], m + 1);
			}
			f16x8 wa[NA];
#pragma unroll
			for (int n = 0; n < NA; ++n) {
				wa[n] = pf_operand<DB>(w[n][m / OPP], m % OPP);
			}
#pragma unroll
			for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
				for (int c = 0; c < NC; ++c) {
#pragma unroll
					for (int n = 0; n < NA; ++n) {
						acc[n][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[n], __builtin_bit_cast(f16x8, bq[m & 1][c * 2 + hl]), acc[n][c], 0, 0, 0);
					}
				}
			}
		}
		// the order asked of the scheduler: A pieces + 4 reads | 4 reads, 8 MFMA | 4 reads, 8 MFMA | 4 reads, 8 MFMA | 8 MFMA
		__builtin_amdgcn_sched_group_barrier(0x100, NA * P + 4, 0);
#pragma unroll
		for (int m = 0; m < 3; ++m) {
			__builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
			__builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
		}
		__builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
	};

#pragma unroll
	for (int d = 0; d < AB; ++d) {
		load_b(fb[d], s_begin + d);
	}
#pragma unroll
	for (int d = 0; d < AA; ++d) {
		load_a(fa[d], s_begin + d);
	}
	stage_b(fb[0], 0);
	__syncthreads();
	// step t of this workgroup's range (absolute step s): fetch B(s + AB) and A(s + AA), stage B(s + 1) (fetched a step ago) and
	// A(s), multiply step s, barrier.  Fetches past the range read real data that is not used.
	const int nloc = s_end - s_begin;
	for (int t0 = 0; t0 < nloc; t0 += U) {
#pragma unroll
		for (int I = 0; I < U; ++I) {
			const int t = t0 + I, s = s_begin + t;
			if (t < nloc) {
				load_b(fb[I % AB], s + AB);
				load_a(fa[(I + AA) % NWB], s + AA);
				__builtin_amdgcn_sched_barrier(0);
				stage_b(fb[(I + 1) % AB], (t + 1) % 3);
				stage_a(fa[I % NWB], s);
				compute(t % 3);
				__syncthreads();
			}
		}
	}
	if (KS > 1) {
		// partial tile out (register order: coalesced both ways), count in; the last one in folds all of them, its own included,
		// in range order -- the sum does not depend on who came last
		__shared__ unsigned arrived;
		const int tile = bx * ny + by;
		float* mine = a.partial + (((size_t)tile * KS + ks) * 4 + wave) * (NA * NC * 16 * 64) + lane;
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					// Written through (agent scope = sc1 stores), then vmcnt(0), then the relaxed agent-scope arrival; the folding
					// workgroup reads with agent-scope (sc1) loads, which are served past its L1: the CDNA guide's "sc1 payload ->
					// drained vmcnt -> sc1 flag" hand-off (MI355X_MICROARCH.md, handoff-flag / valid forms).  That is a property of
					// gfx942 / gfx950's write-through and cache-bypass behaviour, not a release/acquire pair of the memory model: a
					// release fence here writes back the XCD's whole L2 (measured 70-400 us per launch).  Stress-tested over
					// repeated launches at 2..8 ranges (tests/test_prefill_gemm.py); the counters are re-zeroed per prefill call.
					__hip_atomic_store(mine + ((n * NC + c) * 16 + r) * 64, acc[n][c][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			arrived = __hip_atomic_fetch_add(a.tile_count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__syncthreads();
		if (arrived != (unsigned)KS - 1) {
			return;
		}
		if (threadIdx.x == 0) {
			__hip_atomic_store(a.tile_count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
		}
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					acc[n][c][r] = 0.f;
				}
			}
		}
		for (int p = 0; p < KS; ++p) {
			const float* src = a.partial + (((size_t)tile * KS + p) * 4 + wave) * (NA * NC * 16 * 64) + lane;
#pragma unroll
			for (int n = 0; n < NA; ++n) {
#pragma unroll
				for (int c = 0; c < NC; ++c) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						acc[n][c][r] += __hip_atomic_load(src + ((n * NC + c) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
			}
		}
	}
	if constexpr (EPI == PF_EPI_FFN_UP) {
		pf_epilogue<KVB, EPI, NA, NC>(a, acc, unit0, tok0, j, kk); // its fragment-major stores are 512 bytes contiguous as they are
	} else {
		if (a.M & 3) { // a vocabulary that is not a multiple of 4: rows are not 16-byte aligned
			pf_epilogue<KVB, EPI, NA, NC>(a, acc, unit0, tok0, j, kk);
		} else { // the B ring is free: every wave has passed the loop's last barrier
			pf_epilogue_rows<KVB, EPI, NA>(a, acc, unit0, tok0, (float*)pfw_lds + wave * (32 * 68));
		}
	}
}

// ---- the big form ---------------------------------------------------------------------------------------------------------
// At the rate k_pf_gemm_wide runs, its 32 KB of operands per 64-column step of a 256-unit x 64-token workgroup are what the L2s deliver
// (profiles/r02_prefill_gemm.txt: 567-599 TFLOP/s with the loop's global loads removed, 385-406 as built).  Where a GEMM has enough
// units to fill the chip with tiles FOUR times the size -- the FFN-up and the classifier -- one 8-wave workgroup per CU takes
// 512 units x 128 tokens: a wave owns 128 units x 64 tokens (4 x 2 accumulator tiles, 128 registers); the two waves of a unit strip
// share ONE LDS image of the strip's 128 weight rows, the four waves of a token half share the B rows.  64 KB from the L2s per step
// for four times the multiply-adds, and 24 ds_read_b128 + 8 ds_write_b128 per 64 MFMAs instead of 20 + 8 per 32.  Both operands are
// fetched a step ahead into registers and staged into 2-slot LDS rings (fp8: A 2 x 40 KB, B 2 x 32 KB), one barrier per step.
// Measured on the fp8 shapes at 1024 tokens (tools/experiments/exp_pfgemm_big.hip, profiles/r04_prefill.txt): FFN-up 414 -> 477
// TFLOP/s, classifier 415 -> 498; ahead from ~5/8 of the CUs covered (384 tokens), behind below that (the launcher decides).
// (Staging two stores behind each MFMA group instead of ahead of the step's first: 476 against 498 -- not kept.)
// Long rows and few units (the FFN-down): K cut into ranges across workgroups exactly as in k_pf_gemm_wide (a.ksplit).
// FFN-up: a strip is 64 hidden units -- rows 0..63 of its image are w1's, 64..127 w3's -- so a workgroup covers 256 of them.
// fp8 and gf4 weights (an fp16 step of A is 128 bytes per row: the two rings would not fit the LDS).
template <int EPI>
struct PfBig {
	static constexpr int UNITS = EPI == PF_EPI_FFN_UP ? 256 : 512, TOKENS = 128;
};
template <int DB>
struct PfBigA {
	static constexpr int BR = 64 * DB / 8;      // bytes of a row per step: 32 (gf4), 64 (fp8)
	static constexpr int PR = BR / 16;          // 16-byte pieces per row
	static constexpr int RS = BR + 16;          // row stride of the image
	static constexpr int STRIP = 128 * RS;
	static constexpr int A_SLOT = 4 * STRIP, B_SLOT = 32 * 1024;
	static constexpr int EPI_BYTES = 8 * 32 * (32 * 4 + 4) * 4; // the store epilogue's eight wave-private images (pf_epilogue_rows)
	static constexpr int LDS_BYTES = 2 * (A_SLOT + B_SLOT) > EPI_BYTES ? 2 * (A_SLOT + B_SLOT) : EPI_BYTES;
};

template <int DB, int EPI>
__global__ __launch_bounds__(512, 1) void k_pf_gemm_big(PfGemmArgs a) {
	static_assert(DB == 8 || DB == 4, "k_pf_gemm_big: fp8 and gf4 weights");
	static_assert(EPI == PF_EPI_FFN_UP || EPI == PF_EPI_STORE || EPI == PF_EPI_RESID, "k_pf_gemm_big: the FFN-up, the plain store, the residual GEMM");
	constexpr int G = Fmt<DB>::G;
	constexpr int P = 32 / G, OPP = G / 8;
	constexpr int NA = 4, NC = 2;
	constexpr int PR = PfBigA<DB>::PR, RS = PfBigA<DB>::RS, RPL = 64 / PR;
	constexpr int NQ = 64 / RPL; // wave-loads of A per wave and step (its 64 rows of the strip)
	constexpr int A_SLOT = PfBigA<DB>::A_SLOT, B_SLOT = PfBigA<DB>::B_SLOT, STRIP = PfBigA<DB>::STRIP;
	extern __shared__ u32x4 pfb_lds[];
	unsigned char* const lds = (unsigned char*)pfb_lds;
	u32x4(*bst)[32][64] = (u32x4(*)[32][64])lds; // [2][token group (4) x MFMA (4) x hi, lo][64]
	unsigned char* const abase = lds + 2 * B_SLOT;

	const int lane = lane_id(), wave = wave_id();
	const int j = lane & 31, kk = lane >> 5;
	const int strip = wave & 3, half = wave >> 2;
	const int ny = a.ncols, KS = a.ksplit; // 128-token columns; ranges of K (one workgroup each, folded by the last to arrive: as k_pf_gemm_wide)
	const int idx = blockIdx.x >> 3;
	const int bx = (blockIdx.x & 7) + 8 * (idx / (ny * KS)), by = idx % ny, ks = (idx / ny) % KS; // the XCD-aware order of k_pf_gemm_wide
	if (bx * PfBig<EPI>::UNITS >= a.M) {
		return;
	}
	const int unit_wg = bx * PfBig<EPI>::UNITS, tok_wg = by * 128;
	size_t expert_off = 0;
	if (a.col_expert) { // a grouped GEMM (mixture of experts): k_pf_route padded every expert's rows to whole 128-row columns (gran 2)
		const int e = a.col_expert[2 * by];
		if (e < 0) {
			return;
		}
		expert_off = (size_t)e * a.expert_stride;
	}
	const size_t row_bytes = (size_t)a.K * DB / 8;
	const int row_pieces = (int)(row_bytes / 16);
	const int nsteps = pf_steps(a.K);
	const int s_begin = (int)((long)ks * nsteps / KS), s_end = (int)((long)(ks + 1) * nsteps / KS); // this workgroup's steps

	// this wave stages rows half * 64 .. + 63 of its strip's image: wave-load q covers rows q * RPL + lane / PR, piece lane % PR
	const int apiece = lane % PR;
	const unsigned char* rowq[NQ];
#pragma unroll
	for (int q = 0; q < NQ; ++q) {
		const int r = half * 64 + q * RPL + lane / PR;
		if constexpr (EPI == PF_EPI_FFN_UP) {
			rowq[q] = (const unsigned char*)(half ? a.w1 : a.w0) + expert_off + (size_t)min(unit_wg + strip * 64 + (r & 63), a.M - 1) * row_bytes;
		} else {
			rowq[q] = (const unsigned char*)a.w0 + expert_off + (size_t)min(unit_wg + strip * 128 + r, a.M - 1) * row_bytes;
		}
	}
	const float4* xg = a.xin + (size_t)(tok_wg >> 5) * nsteps * 512 + lane;

	u32x4 fa[NQ], fb[4];
	auto load = [&](int sc) { // clamped, never branched on
		const int scc = min(sc, nsteps - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int row = wave * 4 + r;
			fb[r] = *(const u32x4*)(xg + ((size_t)(row >> 3) * nsteps + scc) * 512 + (row & 7) * 64);
		}
		const int piece = min(scc * PR + apiece, row_pieces - 1);
#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			fa[q] = __builtin_nontemporal_load((gptr16)rowq[q] + piece);
		}
	};
	auto stage = [&](int slot, int sc) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			bst[slot][wave * 4 + r][lane] = fb[r];
		}
		unsigned char* img = abase + slot * A_SLOT + strip * STRIP;
		const bool valid = sc * PR + apiece < row_pieces; // ragged rows: pieces past the row's end multiply as zeros
#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			*(u32x4*)(img + (half * 64 + q * RPL + lane / PR) * RS + apiece * 16) = valid ? fa[q] : (u32x4){0u, 0u, 0u, 0u};
		}
	};

	f32x16 acc[NA][NC];
#pragma unroll
	for (int n = 0; n < NA; ++n) {
#pragma unroll
		for (int c = 0; c < NC; ++c) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				acc[n][c][r] = 0.f;
			}
		}
	}
	// step s out of ring slot SLOT (a compile-time constant: the loop is unrolled by two, so the stores into the other slot are
	// visibly disjoint from this step's reads)
	auto step = [&](auto SLOT, int s) {
		constexpr int slot = decltype(SLOT)::value;
		if (s + 1 < s_end) {
			stage(slot ^ 1, s + 1); // fetched during the step before
		}
		load(s + 2);
		__builtin_amdgcn_sched_barrier(0);
		const unsigned char* img = abase + slot * A_SLOT + strip * STRIP;
		u32x4 w[NA][P];
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int i = 0; i < P; ++i) {
				w[n][i] = *(const u32x4*)(img + (32 * n + j) * RS + (kk * P + i) * 16);
			}
		}
		// B operands in halves of an MFMA group -- the two token tiles of (m, hi) or (m, lo) -- asked for one half-group (8 MFMAs, 256
		// matrix-core cycles: an LDS round trip) ahead: a double buffer of 2 x 2 operands, not 2 x 4 (the kernel sits at the 256-register
		// budget of two waves per SIMD)
		u32x4 bq[2][NC];
		auto read_b = [&](u32x4(&q)[NC], int g) { // g = 2 m + (hi: 0, lo: 1)
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				q[c] = bst[slot][(half * 2 + c) * 8 + g][lane];
			}
		};
		read_b(bq[0], 0);
		f16x8 wa[NA];
#pragma unroll
		for (int g = 0; g < 8; ++g) {
			if (g + 1 < 8) {
				read_b(bq[(g + 1) & 1], g + 1);
			}
			if ((g & 1) == 0) {
#pragma unroll
				for (int n = 0; n < NA; ++n) {
					wa[n] = pf_operand<DB>(w[n][(g >> 1) / OPP], (g >> 1) % OPP);
				}
			}
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int n = 0; n < NA; ++n) {
					acc[n][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[n], __builtin_bit_cast(f16x8, bq[g & 1][c]), acc[n][c], 0, 0, 0);
				}
			}
		}
		// A pieces + 2 reads | 2 reads, 8 MFMA | ... (seven times) | 8 MFMA
		__builtin_amdgcn_sched_group_barrier(0x100, NA * P + NC, 0);
#pragma unroll
		for (int g = 0; g < 7; ++g) {
			__builtin_amdgcn_sched_group_barrier(0x100, NC, 0);
			__builtin_amdgcn_sched_group_barrier(0x008, NA * NC, 0);
		}
		__builtin_amdgcn_sched_group_barrier(0x008, NA * NC, 0);
		__syncthreads();
	};

	load(s_begin);
	stage(0, s_begin);
	load(s_begin + 1);
	__syncthreads();
	for (int s = s_begin; s < s_end; s += 2) {
		step(std::integral_constant<int, 0>(), s);
		if (s + 1 < s_end) {
			step(std::integral_constant<int, 1>(), s + 1);
		}
	}
	if (KS > 1) {
		// Partial tile out, count in; the last workgroup of the tile to arrive folds all of them, its own included, in range order:
		// the hand-off of k_pf_gemm_wide (write-through stores, drained, one relaxed agent-scope arrival; agent-scope loads on the
		// folding side -- see the comment there), 256 KB per workgroup here
		__shared__ unsigned arrived;
		constexpr int PART = NA * NC * 16 * 64; // floats per wave
		const int tile = bx * ny + by;
		float* mine = a.partial + (((size_t)tile * KS + ks) * 8 + wave) * PART + lane;
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					__hip_atomic_store(mine + ((n * NC + c) * 16 + r) * 64, acc[n][c][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			arrived = __hip_atomic_fetch_add(a.tile_count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__syncthreads();
		if (arrived != (unsigned)KS - 1) {
			return;
		}
		if (threadIdx.x == 0) {
			__hip_atomic_store(a.tile_count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
		}
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					acc[n][c][r] = 0.f;
				}
			}
		}
		for (int p = 0; p < KS; ++p) {
			const float* src = a.partial + (((size_t)tile * KS + p) * 8 + wave) * PART + lane;
#pragma unroll
			for (int n = 0; n < NA; ++n) {
#pragma unroll
				for (int c = 0; c < NC; ++c) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						acc[n][c][r] += __hip_atomic_load(src + ((n * NC + c) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
			}
		}
	}

	const int tok0 = tok_wg + half * 64;
	if constexpr (EPI == PF_EPI_FFN_UP) {
		// accumulators n and n + 2 are w1 and w3 of hidden units unit0 + 32 n ..; the fragment-major stores are contiguous as they are
		const int unit0 = unit_wg + strip * 64, hsteps = pf_steps(a.M);
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int token = tok0 + 32 * c + j;
			if (token >= a.nb) {
				continue;
			}
#pragma unroll
			for (int n = 0; n < 2; ++n) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					const int ub = unit0 + 32 * n + 8 * g + 4 * kk;
					if (ub >= a.M) {
						continue;
					}
					float h[4];
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const float up = acc[n][c][4 * g + e], gt = acc[n + 2][c][4 * g + e];
						h[e] = (a.gelu ? act_gelu(up) : act_silu(up)) * gt; // src/infer.c:440-450
					}
					pf_store4(a.out, token, ub, hsteps, h);
				}
			}
		}
	} else {
		const int unit0 = unit_wg + strip * 128;
		if (a.M & 3) { // a vocabulary that is not a multiple of 4: rows are not 16-byte aligned
			pf_epilogue<16, EPI, NA, NC>(a, acc, unit0, tok0, j, kk);
		} else { // the rings are free: every wave has passed the loop's last barrier
			pf_epilogue_rows<16, EPI, NA>(a, acc, unit0, tok0, (float*)lds + wave * (32 * (32 * NA + 4)));
		}
	}
}

} // namespace calm
