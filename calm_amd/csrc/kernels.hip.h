// kernels.hip.h -- gfx950 (MI355X, wave64) device code of the calm decode step.
//
// One decode step = a chain of bandwidth-bound dequant-matvecs over every active weight byte
// (2 FLOP per weight byte at fp8: HBM-bound, no MFMA) plus a handful of tiny vector ops.  The
// functional spec is the reference CPU path src/infer.c:311-472; citations below are into the
// reference tree.  Shape of the solution:
//
//   * one WAVE (64 lanes) per output row (or row group), 16 bytes per lane per load
//     (global_load_dwordx4, 1 KiB per wave-instruction, non-temporal: weights are read once);
//   * the activation vector lives in LDS as fp32 in a lane-interleaved ("swizzled") layout so
//     that every ds_read_b128 of a wave touches 64 consecutive 16-byte slots (conflict-free);
//   * each workgroup issues its first weight loads BEFORE it builds the activation vector
//     (norm prologue), so HBM is streaming while the prologue runs;
//   * fp32 accumulation, wave-shuffle reduction, fused epilogues (bias/clip/RoPE/KV-append,
//     residual add, gated activation, MoE weighting);
//   * per-token scalars (token, pos, kv slot) live in a device-resident TokState so the whole
//     step can be replayed from a hipGraph.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace calm {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// explicit global address space: a pointer selected between two provenances (weights / dummy) would
// otherwise degrade to FLAT loads, which tick lgkmcnt as well as vmcnt and serialise against LDS
typedef const __attribute__((address_space(1))) u32x4* gptr16;

#ifdef CALM_TIMELINE
// tools/timeline.py builds a copy of the library with this macro: every row-engine kernel stamps, per wave, the 100 MHz wall
// clock at entry, when the LDS image is built, after its first tile and at exit into `calm_tl_buf` ([wave][8]; the last launch
// wins).  Not compiled into the product.
__device__ unsigned long long* calm_tl_buf;
__device__ unsigned calm_tl_waves;
#endif

// per-token scalars, written by k_begin_token, read by every other kernel
struct TokState {
	int token;
	int pos;
	int kv_sink; // 0, or CALM_KV_SINKS once pos >= seq_len   (src/infer.c:330)
	int kv_pos;  // cache slot the new K/V row goes to          (src/infer.c:331)
	int kv_len;  // number of valid cache rows                   (src/infer.c:332)
	// decode steps this model has begun: only ever incremented (k_begin_token), never reset -- the tag of the in-launch hand-off
	// granules of k_qkv_attn, which must never match what an earlier step left in the buffer
	unsigned epoch;
	int pad[2];
};

// head size for which prepare_hip keeps the transposed value cache (behind the [position][dim] one, same size)
__host__ __device__ constexpr bool attn_has_vt(int head_dim) {
	return head_dim == 128;
}
// Its layout: [kv head][block of VT_BLOCK_BYTES / ebytes positions][dim][position in the block] -- transposed within blocks of 32
// (binary16) / 64 (e5m2) positions, so that the V^T operand rows of one tile of keys (64 bytes of every dim) are ONE contiguous 8 KiB
// and its wave-loads are as coalesced as those of the K rows.  Element offset of (row = kv head * head_dim + dim, position):
constexpr int VT_BLOCK_BYTES = 64;
__host__ __device__ inline size_t attn_vt_offset(int row, int pos, int head_dim, int seq_len, int ebytes) {
	const int pb = VT_BLOCK_BYTES / ebytes; // positions per block
	const int kvh = row / head_dim, d = row % head_dim;
	return (((size_t)kvh * (seq_len / pb) + pos / pb) * head_dim + d) * pb + pos % pb;
}

// weights per 16-byte lane-load and float4s of activation it pairs with
template <int DB>
struct Fmt {
	static constexpr int G = 128 / DB; // fp16: 8, fp8: 16, gf4: 32
	static constexpr int F4 = G / 4;
	// The LDS activation image, per 1-KiB chunk of a row: F4 ROWS of float4 slots -- row i holds "float4 #i" of each of the 64 lanes --
	// and for gf4 one more row holding the activation sum of each of a lane's four 8-weight words (see dot16<4>).  A row is 64 slots
	// plus PAD: the staging threads write consecutive LOGICAL float4s, i.e. F4 rows at once, and with rows exactly 1 KiB apart the
	// 8 lanes of one LDS pass (128 bytes) all fell into the same banks -- 32 cycles per 1-KiB store at fp8, 64 at gf4, where 8 would do
	// (SQ_LDS_BANK_CONFLICT: 24 / 56 cycles per store instruction, profiles/r03_pmc_gf4_tables.txt).  PAD = 8 / F4 slots shifts row i by
	// i x the bytes its lanes of a pass cover, so a pass covers 128 consecutive bank bytes; reads (a wave reads one row) stay contiguous.
	static constexpr int PAD = 8 / F4;
	static constexpr int ROW = 64 + PAD;
	static constexpr int CS = ROW * (F4 + (DB == 4 ? 1 : 0));
};

__device__ __forceinline__ int lane_id() {
	return threadIdx.x & 63;
}
__device__ __forceinline__ int wave_id() {
	// wave-uniform by construction; tell the compiler so it lands in an SGPR
	return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// Sum over the 64 lanes with DPP row operations (VALU only; no LDS crossbar round trips):
// quad swaps, row_shr:4, row_shr:8 leave each 16-lane row's total in its lanes 12..15, row_bcast:15
// and row_bcast:31 carry the totals across rows.  The full sum is valid in LANE 63 only.
constexpr int RED_LANE = 63;
__device__ __forceinline__ float wave_sum63(float v) {
	auto dpp = [](float x, auto ctrl, auto row_mask) {
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, true));
	};
	using I = std::integral_constant<int, 0>;
	(void)sizeof(I);
	v += dpp(v, std::integral_constant<int, 0xb1>(), std::integral_constant<int, 0xf>());  // quad_perm [1,0,3,2]
	v += dpp(v, std::integral_constant<int, 0x4e>(), std::integral_constant<int, 0xf>());  // quad_perm [2,3,0,1]
	v += dpp(v, std::integral_constant<int, 0x114>(), std::integral_constant<int, 0xf>()); // row_shr:4
	v += dpp(v, std::integral_constant<int, 0x118>(), std::integral_constant<int, 0xf>()); // row_shr:8
	v += dpp(v, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>()); // row_bcast:15 into rows 1,3
	v += dpp(v, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>()); // row_bcast:31 into rows 2,3
	return v;
}
// same, broadcast to every lane (one v_readlane)
__device__ __forceinline__ float wave_sum(float v) {
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum63(v)), RED_LANE));
}
// Sum over each group of LPR consecutive lanes (LPR = 4 .. 64, a power of two), result in EVERY lane of the
// group.  Up to 16 lanes (one DPP row) it is VALU only: quad swaps, then row_half_mirror and row_mirror add
// the other 4 / 8 lanes' (already uniform) sums; wider groups finish with cross-row shuffles.  (__shfl_xor
// compiles to ds_bpermute + s_waitcnt: the attention kernels spent more time there than in their loads.)
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
	auto dpp = [](float x, auto ctrl) {
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
	};
	if constexpr (LPR >= 2) {
		v += dpp(v, std::integral_constant<int, 0xb1>()); // quad_perm [1,0,3,2]
	}
	if constexpr (LPR >= 4) {
		v += dpp(v, std::integral_constant<int, 0x4e>()); // quad_perm [2,3,0,1]
	}
	if constexpr (LPR >= 8) {
		v += dpp(v, std::integral_constant<int, 0x141>()); // row_half_mirror
	}
	if constexpr (LPR >= 16) {
		v += dpp(v, std::integral_constant<int, 0x140>()); // row_mirror
	}
#pragma unroll
	for (int ofs = 16; ofs < LPR; ofs <<= 1) {
		v += __shfl_xor(v, ofs);
	}
	return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		v = fmaxf(v, __shfl_xor(v, o));
	}
	return v;
}

// group_sum of N independent values, stage by stage: every DPP add of a stage reads a register written N - 1 instructions earlier
// (one chain per value gets two wait states in front of each of its four steps)
template <int LPR, int N>
__device__ __forceinline__ void group_sum_multi(float (&v)[N]) {
	auto dpp = [](float x, auto ctrl) {
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
	};
	static_assert(LPR <= 16, "group_sum_multi: groups within one DPP row");
	if constexpr (LPR >= 2) {
#pragma unroll
		for (int i = 0; i < N; ++i) {
			v[i] += dpp(v[i], std::integral_constant<int, 0xb1>());
		}
	}
	if constexpr (LPR >= 4) {
#pragma unroll
		for (int i = 0; i < N; ++i) {
			v[i] += dpp(v[i], std::integral_constant<int, 0x4e>());
		}
	}
	if constexpr (LPR >= 8) {
#pragma unroll
		for (int i = 0; i < N; ++i) {
			v[i] += dpp(v[i], std::integral_constant<int, 0x141>());
		}
	}
	if constexpr (LPR >= 16) {
#pragma unroll
		for (int i = 0; i < N; ++i) {
			v[i] += dpp(v[i], std::integral_constant<int, 0x140>());
		}
	}
}
// maximum over the 64 lanes, in every lane (VALU + one v_readlane; wave_max above goes through the LDS crossbar six times)
__device__ __forceinline__ float wave_max_dpp(float v) {
	auto dpp = [](float x, auto ctrl, auto row_mask) { // lanes the control leaves without a source keep x
		const int xi = __builtin_bit_cast(int, x);
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
	};
	using F = std::integral_constant<int, 0xf>;
	v = fmaxf(v, dpp(v, std::integral_constant<int, 0xb1>(), F()));
	v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4e>(), F()));
	v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>(), F()));
	v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>(), F()));
	v = fmaxf(v, dpp(v, std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>())); // row_bcast:15 into rows 1, 3
	v = fmaxf(v, dpp(v, std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>())); // row_bcast:31 into rows 2, 3
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), RED_LANE));
}

// ---------------------------------------------------------------- weight decode ---------------

// fp8 e5m2 == OCP bf8 on gfx950; exact.  (reference src/infer.c:28-35: bits << 8 as binary16)
__device__ __forceinline__ f32x2 bf8x2_lo(unsigned w) {
	return __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false);
}
__device__ __forceinline__ f32x2 bf8x2_hi(unsigned w) {
	return __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
}
__device__ __forceinline__ float bf8_byte0(unsigned w) {
	return __builtin_amdgcn_cvt_f32_bf8((int)w, 0);
}

// two fp32 values -> two fp8 e5m2 bytes (low byte = a), one round-to-nearest-even step each, overflow and infinity saturating
// to the largest finite code (57344): what `__nv_fp8_e5m2(float)` does for the reference's fp8 KV rows (src/infer.cu:473-482)
// and what oracle_float_to_e5m2 restates.
__device__ __forceinline__ unsigned short e5m2x2_sat(float a, float b) {
	a = a > 57344.0f ? 57344.0f : (a < -57344.0f ? -57344.0f : a); // unordered compares are false: NaN stays NaN
	b = b > 57344.0f ? 57344.0f : (b < -57344.0f ? -57344.0f : b);
	return (unsigned short)(__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffff);
}

__device__ __forceinline__ float half_bits_to_float(unsigned short h) {
	return __half2float(__ushort_as_half(h));
}

// scalar decode of element idx of a weight tensor (embedding gather; src/infer.c:334-347)
template <int DB>
__device__ __forceinline__ float decode_elem(const void* w, size_t idx) {
	if constexpr (DB == 16) {
		return half_bits_to_float(((const unsigned short*)w)[idx]);
	} else if constexpr (DB == 8) {
		unsigned b = ((const unsigned char*)w)[idx];
		return bf8_byte0(b);
	} else {
		unsigned v = ((const unsigned*)w)[idx >> 3];
		float s = bf8_byte0(v) * -0.25f; // src/infer.c:38
		int q = (int)((v >> (8 + 3 * (idx & 7))) & 7) - 4;
		return (float)q * s;
	}
}

// acc + (binary16 in the low / high half of h2) * x, fp32 result: one VALU instruction, no conversion
__device__ __forceinline__ float fma_mix_lo(unsigned h2, float x, float acc) {
	asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h2), "v"(x));
	return acc;
}
__device__ __forceinline__ float mul_mix_lo(unsigned h2, float x) {
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(x));
	return r;
}
__device__ __forceinline__ float mul_mix_hi(unsigned h2, float x) {
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(x));
	return r;
}
__device__ __forceinline__ float fma_mix_hi(unsigned h2, float x, float acc) {
	asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h2), "v"(x));
	return acc;
}

// acc += dot(16-byte lane-load `v`, its G activations), as a 2-wide accumulator (even/odd columns)
// so that every multiply-add is a v_pk_fma_f32 on register pairs that are already adjacent: the
// converted weight pair, the activation pair from ds_read_b128, the accumulator pair.
// xp points at this lane's first float4 of the chunk in the swizzled LDS image; float4 #i of the
// lane is at xp[i * ROW] (ROW = Fmt<DB>::ROW slots).
// X(i): the lane's float4 #i of the chunk's activations -- from the LDS image (dot16) or from registers (run_rows_impl XR)
template <int DB, class XF>
__device__ __forceinline__ f32x2 dot16x(u32x4 v, XF X, f32x2 acc) {
	// (fp16 / fp8: a second, call-local accumulator pair halves the length of the dependent v_pk_fma chain
	// -- PMC showed 30-40 % of wave cycles in issue stalls with one chain per row)
	if constexpr (DB == 16) {
		f32x2 acc_b = {0.f, 0.f};
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			f32x4 x = X(i);
			unsigned w0 = v[2 * i], w1 = v[2 * i + 1];
			f32x2 a = {half_bits_to_float((unsigned short)(w0 & 0xffff)), half_bits_to_float((unsigned short)(w0 >> 16))};
			f32x2 b = {half_bits_to_float((unsigned short)(w1 & 0xffff)), half_bits_to_float((unsigned short)(w1 >> 16))};
			acc = __builtin_elementwise_fma(a, x.lo, acc);
			acc_b = __builtin_elementwise_fma(b, x.hi, acc_b);
		}
		acc += acc_b;
	} else if constexpr (DB == 8) {
		f32x2 acc_b = {0.f, 0.f};
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			f32x4 x = X(i);
			acc = __builtin_elementwise_fma(bf8x2_lo(v[i]), x.lo, acc);
			acc_b = __builtin_elementwise_fma(bf8x2_hi(v[i]), x.hi, acc_b);
		}
		acc += acc_b;
	} else {
		// gf4: word = 8-bit e5m2 scale S + 8 x 3-bit codes, w_k = (q_k - 4) * S / -4   (src/infer.c:37-40)
		//   sum_k w_k x_k = (-S/4) * sum_k q_k x_k + S * sum_k x_k
		// The codes are never converted.  A 3-bit field anywhere in the mantissa of a binary16 half, everything
		// else masked off, IS the subnormal q * 2^(a-24), and v_fma_mix_f32 multiplies a half by an fp32
		// activation into an fp32 accumulator in one instruction; the LDS image carries x_k * 2^-a per column
		// (exact), so the product is q x 2^-24 whatever a is.  One rotation of the word puts c5 c6 c7 into the
		// mantissa of the low half and c0 c1 c2 into that of the high half, so three ANDs isolate six codes;
		// c3 and c4 already sit in the high half's mantissa of the word itself: 6 integer ops + 8 fma_mix per
		// 8 weights, and the activation sum of the word comes precomputed from the image (float4 #8).
		// (Round 4: the chain started FROM -2^-22 x that sum, one multiply-add fewer per word and the factor -2^22 applied once per row,
		// measured the same to the tenth of a microsecond on every gf4 kernel -- profiles/r04_gf4.txt: the kernels are not bound by
		// their VALU count.  Not kept.)
		f32x4 xv[4][2];
		unsigned m[4][5];
		float t[4], S[4];
		const f32x4 xsum = X(8);
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const unsigned w = v[j];
			xv[j][0] = X(2 * j);
			xv[j][1] = X(2 * j + 1);
			S[j] = bf8_byte0(w);
			const unsigned r = __builtin_amdgcn_alignbit(w, w, 23); // rotate right by 23
			m[j][0] = r & 0x000E0007u; // lo: c5 (a = 0)   hi: c0 (a = 1)
			m[j][1] = r & 0x00700038u; // lo: c6 (a = 3)   hi: c1 (a = 4)
			m[j][2] = r & 0x038001C0u; // lo: c7 (a = 6)   hi: c2 (a = 7)
			m[j][3] = w & 0x000E0000u; //                  hi: c3 (a = 1)
			m[j][4] = w & 0x00700000u; //                  hi: c4 (a = 4)
		}
		// v_fma_mix_f32 is written as (pure, non-volatile) inline asm: left to itself the SLP vectoriser
		// keeps re-pairing these chains into v_pk_fma_f32 and converts every code with v_cvt_f32_f16 again.
		// One chain per word; the four words' chains are interleaved code-major for ILP.
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			t[j] = mul_mix_hi(m[j][0], xv[j][0][0]);
		}
#pragma unroll
		for (int k = 1; k < 8; ++k) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float x = xv[j][k >> 2][k & 3];
				t[j] = k < 5 ? fma_mix_hi(m[j][k], x, t[j]) : fma_mix_lo(m[j][k - 5], x, t[j]);
			}
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			// (-S/4) * 2^24 * t + S * xsum  =  S * (xsum - 2^22 t): two fmas on one chain
			acc[0] = fmaf(S[j], fmaf(t[j], -4194304.0f, xsum[j]), acc[0]);
		}
	}
	return acc;
}

template <int DB>
__device__ __forceinline__ f32x2 dot16(u32x4 v, const f32x4* xp, f32x2 acc) {
	return dot16x<DB>(v, [&](int i) { return xp[i * Fmt<DB>::ROW]; }, acc);
}

// ---------------------------------------------------------------- activation staging ----------

// Swizzled LDS image of an n-float vector for weight format DB.  Logical float4 p (columns
// 4p..4p+3) belongs to chunk p / (16G), lane (p % 16G) / F4, sub-index i = p % F4 and is stored at
// float4 slot chunk*CS + i*ROW + lane (Fmt<DB>: ROW = 64 + PAD): a wave reading "its float4 #i" hits 64 consecutive slots.
// gf4 only: the image holds x_k * 2^-a(k % 8), a = {1,4,7,1,4,0,3,6} (dot16<4> multiplies by codes that
// sit 2^a too high), and slot chunk*CS + 8*ROW + lane holds the four UNSCALED 8-column sums of the lane.
template <int DB>
__device__ __forceinline__ int swz4(int p) {
	constexpr int G = Fmt<DB>::G, F4 = Fmt<DB>::F4;
	int chunk = p / (16 * G), r = p % (16 * G);
	return chunk * Fmt<DB>::CS + (r % F4) * Fmt<DB>::ROW + r / F4;
}

// logical float4 count of the image of n floats (whole chunks; the tail is zero-filled) ...
template <int DB>
__host__ __device__ constexpr int xs_logical(int n) {
	return ((n + 64 * Fmt<DB>::G - 1) / (64 * Fmt<DB>::G)) * 16 * Fmt<DB>::G;
}
// ... and the float4 slots it occupies in LDS
template <int DB>
__host__ __device__ constexpr int xs_slots(int n) {
	return ((n + 64 * Fmt<DB>::G - 1) / (64 * Fmt<DB>::G)) * Fmt<DB>::CS;
}

// gf4: write float4 p of the image (scaled per column) and, from the even p of a pair, the pair's unscaled sum.
// Lanes p and p^1 are adjacent threads (p = tid + i * BLOCK, BLOCK even) and are active together (n % 32 == 0).
__device__ __forceinline__ void stage_store_gf4(float4* xs4, int p, float4 t) {
	float s = (t.x + t.y) + (t.z + t.w);
	s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xb1, 0xf, 0xf, true)); // lane ^ 1 (quad_perm [1,0,3,2])
	const bool odd = p & 1;
	t.x *= odd ? 0.0625f : 0.5f;       // 2^-a: columns 4..7 = {4,0,3,6}, columns 0..3 = {1,4,7,1}
	t.y *= odd ? 1.0f : 0.0625f;
	t.z *= odd ? 0.125f : 0.0078125f;
	t.w *= odd ? 0.015625f : 0.5f;
	xs4[swz4<4>(p)] = t;
	if (!odd) {
		const int chunk = p / 512, r = p % 512; // 512 logical float4 per gf4 chunk; lane r / 8, word (r % 8) / 2
		((float*)&xs4[chunk * Fmt<4>::CS + 8 * Fmt<4>::ROW + r / 8])[(r % 8) >> 1] = s;
	}
}

// block-wide sum; every thread gets the same value.  red: >= BLOCK/64 floats of LDS.
template <int BLOCK>
__device__ __forceinline__ float block_sum(float v, float* red) {
	constexpr int NW = BLOCK / 64;
	v = wave_sum(v);
	__syncthreads(); // protect red from the previous use
	if (lane_id() == 0) {
		red[threadIdx.x >> 6] = v;
	}
	__syncthreads();
	float s = 0.f;
#pragma unroll
	for (int i = 0; i < NW; ++i) {
		s += red[i];
	}
	return s;
}

// Staging an n-float vector into the swizzled image happens in two halves so that a kernel can issue
// the vector's global loads BEFORE its first weight tile (vmcnt retires in order: the prologue must
// not have to wait for the weight loads queued behind it) and finish after:
//   stage_load   : issue the loads of src into V float4 registers per thread
//   stage_finish : normalise (optional), write the LDS image, barrier
//     normw == nullptr : plain copy
//     else             : (x - mean) * rsqrt(var + eps) * normw, mean = 0 unless ln   (src/infer.c:183-207)
// If dump != nullptr, block 0 also writes the staged (unswizzled) vector there (norm_par models
// need the attention-norm output again in the FFN, src/infer.c:417-420).
// Every load is UNCONDITIONAL and unclamped -- it may read up to 4*V*BLOCK floats regardless of n;
// the surplus is masked where it is used.  (A load behind a branch makes the number of outstanding
// loads unknowable to the compiler, which then waits with vmcnt(0), i.e. for the weight tile too;
// a clamped index costs a 64-bit address per load.)  Every device buffer is therefore allocated
// with DEV_PAD bytes of slack (infer_hip.hip).  Vectors longer than 4*V*BLOCK floats fall back to
// re-reading global memory for the excess.
// NORM: the kernel may have a norm weight to apply (then its loads are issued here too; a null normw at
// run time -- parallel-residual models -- re-reads src instead, unconditionally all the same).
template <int V, bool NORM>
struct StageRegs {
	float4 v[V];
	float4 g[NORM ? V : 1];
};

template <int BLOCK, int V, bool NORM>
__device__ __forceinline__ void stage_load(StageRegs<V, NORM>& sr, const float* __restrict__ src, const float* __restrict__ normw) {
	const float4* src4 = (const float4*)src + threadIdx.x;
#pragma unroll
	for (int i = 0; i < V; ++i) {
		sr.v[i] = src4[i * BLOCK];
	}
	if constexpr (NORM) {
		const float4* g4 = (const float4*)(normw ? normw : src) + threadIdx.x;
#pragma unroll
		for (int i = 0; i < V; ++i) {
			sr.g[i] = g4[i * BLOCK];
		}
	}
}

template <int DB, int BLOCK, int V, bool NORM>
// Returns the factor the caller's epilogue still has to apply to every dot product with the image: 1, except for RMSNorm
// (normw, !ln, no dump), where the image holds x * normw and the factor is rsqrt(mean(x^2) + eps) -- a scalar commutes with the
// matvec, so the image does not wait for the block-wide sum of squares (two barriers and an LDS round trip on the path to the first
// multiply-add); the sum rides on the image's own barrier instead.
__device__ __forceinline__ float stage_finish(const StageRegs<V, NORM>& sr, float4* xs4, float* red, const float* __restrict__ src, const float* __restrict__ normw,
                                              int n, float eps, bool ln, float* dump, int dump_block = 0) {
	constexpr int MAXV = V;
	constexpr int NWAVES = BLOCK / 64;
	const int tid = threadIdx.x;
	const int n4 = n >> 2;
	const int slots = xs_logical<DB>(n);
	const float4* src4 = (const float4*)src;
	const float4(&v)[MAXV] = sr.v;

	const bool deferred = normw && !ln && !dump;
	float mean = 0.f, scale = 1.f;
	if (deferred) {
		float ss = 0.f;
		auto put = [&](int p, float4 t, float4 gw) {
			ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
			t.x *= gw.x, t.y *= gw.y, t.z *= gw.z, t.w *= gw.w;
			if constexpr (DB == 4) {
				stage_store_gf4(xs4, p, t);
			} else {
				xs4[swz4<DB>(p)] = t;
			}
		};
#pragma unroll
		for (int i = 0; i < MAXV; ++i) {
			int p = tid + i * BLOCK;
			if (p < n4) {
				put(p, v[i], sr.g[NORM ? i : 0]);
			}
		}
		for (int p = tid + MAXV * BLOCK; p < n4; p += BLOCK) {
			put(p, src4[p], ((const float4*)normw)[p]);
		}
		ss = wave_sum(ss);
		if (lane_id() == 0) {
			red[tid >> 6] = ss;
		}
	} else if (normw) {
		if (ln) {
			float s = 0.f;
#pragma unroll
			for (int i = 0; i < MAXV; ++i) {
				if (tid + i * BLOCK < n4) {
					s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
				}
			}
			for (int p = tid + MAXV * BLOCK; p < n4; p += BLOCK) {
				float4 t = src4[p];
				s += (t.x + t.y) + (t.z + t.w);
			}
			mean = block_sum<BLOCK>(s, red) / (float)n;
		}
		float ss = 0.f;
#pragma unroll
		for (int i = 0; i < MAXV; ++i) {
			int p = tid + i * BLOCK;
			if (p < n4) {
				float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
				ss += (a * a + b * b) + (c * c + d * d);
			}
		}
		for (int p = tid + MAXV * BLOCK; p < n4; p += BLOCK) {
			float4 t = src4[p];
			float a = t.x - mean, b = t.y - mean, c = t.z - mean, d = t.w - mean;
			ss += (a * a + b * b) + (c * c + d * d);
		}
		float var = block_sum<BLOCK>(ss, red) / (float)n;
		scale = 1.0f / sqrtf(var + eps);
	}

	auto emit = [&](int p, float4 t, float4 gw) {
		if (normw) {
			t.x = (t.x - mean) * scale * gw.x;
			t.y = (t.y - mean) * scale * gw.y;
			t.z = (t.z - mean) * scale * gw.z;
			t.w = (t.w - mean) * scale * gw.w;
		}
		if (dump && (int)blockIdx.x == dump_block) {
			((float4*)dump)[p] = t;
		}
		if constexpr (DB == 4) {
			stage_store_gf4(xs4, p, t);
		} else {
			xs4[swz4<DB>(p)] = t;
		}
	};
	if (!deferred) {
#pragma unroll
		for (int i = 0; i < MAXV; ++i) {
			int p = tid + i * BLOCK;
			if (p < n4) {
				emit(p, v[i], sr.g[NORM ? i : 0]);
			}
		}
		for (int p = tid + MAXV * BLOCK; p < n4; p += BLOCK) {
			emit(p, src4[p], normw ? ((const float4*)normw)[p] : make_float4(0.f, 0.f, 0.f, 0.f));
		}
	}
	// zero the tail of the last chunk so masked-off lanes multiply 0 * 0
	for (int p = n4 + tid; p < slots; p += BLOCK) {
		if constexpr (DB == 4) {
			stage_store_gf4(xs4, p, make_float4(0.f, 0.f, 0.f, 0.f));
		} else {
			xs4[swz4<DB>(p)] = make_float4(0.f, 0.f, 0.f, 0.f);
		}
	}
	__syncthreads();
	if (deferred) {
		float ss = 0.f;
#pragma unroll
		for (int i = 0; i < NWAVES; ++i) {
			ss += red[i];
		}
		return 1.0f / sqrtf(ss / (float)n + eps);
	}
	return 1.0f;
}

// ---------------------------------------------------------------- the row engine --------------

// A tile = U consecutive 1-KiB wave-loads of each of NR rows, held in registers.
template <int NR, int U>
struct Tile {
	u32x4 w[U][NR];
};

// FULL: every row is a whole number of 1-KiB wave-loads (nl % 64 == 0), so no lane is ever masked.
// Loads are always issued.  Surplus chunks of a row's last step are clamped to the row's last chunk (a cache hit instead of the
// next row's first KiBs from HBM a second time: DBRX's 10.5-KiB w2 rows walked 4 chunks at a time asked for 12 KiB each);
// a ragged row's (!FULL) last chunk runs up to 1 KiB past the row's end (into the next row, or into the DEV_PAD
// slack behind the tensor) and tile_fma zeroes that data.
template <int DB, int NR, int U, bool FULL>
__device__ __forceinline__ void tile_load(Tile<NR, U>& t, const unsigned char* const (&rows)[NR], int k0, int nl, int lane) {
#pragma unroll
	for (int u = 0; u < U; ++u) {
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			// clamp the (wave-uniform) chunk index into the row -- a surplus chunk re-reads the row's
			// last KiB (cache hit) instead of pulling the next row's first KiB from HBM a second time
			const int c = min(k0 + u, ((nl + 63) >> 6) - 1);
			t.w[u][r] = __builtin_nontemporal_load((gptr16)rows[r] + c * 64 + lane);
		}
	}
}

template <int DB, int NR, int U, bool FULL>
__device__ __forceinline__ void tile_fma(const Tile<NR, U>& t, f32x2 (&acc)[NR], const float4* xs4, int k0, int nl, int lane) {
#pragma unroll
	for (int u = 0; u < U; ++u) {
		if ((k0 + u) * 64 < nl) { // wave-uniform: skip chunks past the end of the row
			const f32x4* xp = (const f32x4*)xs4 + (k0 + u) * Fmt<DB>::CS + lane;
			const bool live = FULL || (k0 + u) * 64 + lane < nl;
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				u32x4 w = t.w[u][r];
				if (!FULL) {
					w = live ? w : (u32x4){0u, 0u, 0u, 0u};
				}
				acc[r] = dot16<DB>(w, xp, acc[r]);
			}
		}
	}
}

// The same with the lane's activations held in REGISTERS (run_rows_impl XR): a lane only ever multiplies with "its" float4s of every
// chunk of the vector -- the same ones for every row -- so where a row is XR chunks long and XR x NF float4s fit the register file
// (dim 4096 at gf4: 2 x 9 float4 = 72 VGPRs) they are read from the LDS image ONCE, after it is built, and no step reads LDS again.
// c0: the (compile-time after unrolling) chunk the step starts at.  Rows are whole chunks (the launcher's condition).
template <int DB>
struct XRegs {
	static constexpr int NF = Fmt<DB>::F4 + (DB == 4 ? 1 : 0);
};
template <int DB, int NR, int U, int XR>
__device__ __forceinline__ void tile_fma_regs(const Tile<NR, U>& t, f32x2 (&acc)[NR], const f32x4 (&xr)[XR > 0 ? XR : 1][XRegs<DB>::NF], int c0) {
#pragma unroll
	for (int u = 0; u < U; ++u) {
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			acc[r] = dot16x<DB>(t.w[u][r], [&](int i) { return xr[c0 + u][i]; }, acc[r]);
		}
	}
}

// Runs `ntasks` row-group tasks over the workgroup's waves.  Wave-task t (t = first, first +
// stride, ...) owns NR rows given by rows_of(t, rows).  stage() builds the LDS activation image and
// must end with a barrier; it is called AFTER the first tile's loads have been issued so that the
// weight stream starts before the prologue.  pre() issues the activation loads ahead of that tile.
// epi(t, acc) runs on every lane; the reduced sums are valid in lane RED_LANE only.
//
// The tile stream is software-pipelined two steps deep across k-steps AND across tasks (see the body),
// so every wave keeps up to 2 x 8 KiB in flight and HBM never waits for a wave's reduction tail.  The two
// register tiles alternate through a 2x-unrolled loop body so all tile indices are compile-time.
// aux_of(t, aux) may issue small loads the epilogue needs (residual value, RoPE pair); it runs before the
// task's last multiply-add so that latency hides behind it.  epi(t, acc, aux): sums valid in lane RED_LANE.
//
// SEGS (k_ffn_down of a mixture of MANY SMALL experts): a task's "row" is the CONCATENATION of `segs` rows of n columns each -- the
// same output row of several experts' matrices, each against its own expert's hidden vector, all of them side by side in LDS -- so
// that one uninterrupted tile stream covers all of them (one pass per expert pays a whole kernel start-up each time: OLMoE's eight
// 2048 x 1024 matrices took 21.6 us for 16.8 MB, profiles/r04_shape_sweep.txt; where the images are large -- Mixtral's 57 KB, DBRX's
// 43 KB -- the longer prologue costs more than the restarts: profiles/r04_moe.txt, and the launcher keeps one pass per expert).  rows_of(t, seg, rows)
// names segment seg's rows; segment seg's image starts seg * (chunks of n) * CS slots into xs4; every segment is walked in whole
// steps (its last step's surplus chunks clamp into the row, as they do at the end of any row); at the end of EVERY segment the
// sums are reduced and handed to epi(t, seg, acc, aux), which keeps the running value -- the experts are added in rank order by
// the same lane, exactly as with one pass per expert.  aux_of runs before segment 0's last multiply-add.
// XR > 0: rows of exactly XR chunks, the lane's activations in registers (tile_fma_regs); a row is then one step (U == XR) or two
// (2 U == XR) long, and because every task has that many steps and the first one starts in phase 0 of the two-phase loop body, the
// phase IS the step within the row: the chunk index is a compile-time constant.
template <int DB, int NR, int U, bool FULL, bool SEGS, int XR, class RowsFn, class PreFn, class StageFn, class AuxFn, class EpiFn>
__device__ __forceinline__ void run_rows_impl(int ntasks, int first, int stride, int n, int segs, const float4* xs4, const void* dummy, RowsFn rows_of, PreFn pre,
                                              StageFn stage, AuxFn aux_of, EpiFn epi) {
	static_assert(XR == 0 || (!SEGS && FULL && (XR == U || XR == 2 * U)), "XR: whole rows of one or two steps");
	const int lane = lane_id();
	const int nl = n / Fmt<DB>::G;
	const int cpi = (nl + 63) >> 6;           // chunks of one segment's row (and of its image)
	const int cps = (cpi + U - 1) / U * U;    // ... walked in whole steps
	const int ksteps = SEGS ? segs * cps : 0; // chunk positions of a task
	const unsigned char* rows[2][NR];
	Tile<NR, U> tile[2];
	f32x2 acc2[NR];
	float acc[NR], aux[NR];

	// The stream of tile steps (task t, k-offset k0) is walked with TWO steps always in flight:
	//   prologue : issue step 0 and step 1, then build the LDS image (stage) while they fly;
	//   step s   : [last step of a task: issue the epilogue's small loads (aux_of)]
	//              multiply-add tile s  ->  re-issue that register tile with step s+2
	//              [last step of a task: reduce, epilogue]
	// NOTHING issues a load conditionally: past the end of a wave's work the "next step" reads
	// `dummy` (a small L2-resident buffer with DEV_PAD slack) and the data is dropped, so the
	// compiler's vmcnt bookkeeping is exact and every wait is "the tile two issues ago".
	auto advance = [&](int& t, int& k0, bool& live) { // -> the step after (t, k0)
		k0 += U;
		if (SEGS ? k0 >= ksteps : k0 * 64 >= nl) {
			k0 = 0;
			t += stride;
			live = live && t < ntasks;
		}
	};
	auto seg_start = [&](int k0) { return SEGS ? k0 % cps == 0 : k0 == 0; }; // (t, k0) opens a row (SEGS: a segment)
	auto issue = [&](int ph, int t, int k0, bool live) {
		if (seg_start(k0) || !live) {
			if constexpr (SEGS) {
				rows_of(min(t, ntasks - 1), live ? k0 / cps : 0, rows[ph]);
			} else {
				rows_of(min(t, ntasks - 1), rows[ph]);
			}
		}
		if (!live) {
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				rows[ph][r] = (const unsigned char*)dummy;
			}
		}
		tile_load<DB, NR, U, FULL>(tile[ph], rows[ph], live ? (SEGS ? k0 % cps : k0) : 0, nl, lane);
	};

#ifdef CALM_TIMELINE
	unsigned long long tl[4];
	tl[0] = wall_clock64(), tl[1] = tl[2] = tl[3] = 0;
	auto tl_flush = [&]() {
		tl[3] = wall_clock64();
		const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
		if (lane == 0 && calm_tl_buf && w < calm_tl_waves) {
			unsigned long long* o = calm_tl_buf + (size_t)w * 8;
			o[0] = tl[0], o[1] = tl[1], o[2] = tl[2], o[3] = tl[3];
		}
	};
#endif
	if (first >= ntasks) {
		// a wave without a single task (more waves than tasks) only helps to build the LDS image: none of the always-issued
		// dummy tile loads below, which would sit in the CU's memory queue ahead of its neighbours' real tiles
		pre();
		stage();
#ifdef CALM_TIMELINE
		tl[1] = wall_clock64();
		tl_flush();
#endif
		return;
	}
	pre(); // the activation vector's loads go first: they must retire before, not behind, the tiles
	int t = first, k0 = 0;   // step being consumed
	bool live = t < ntasks;
	int t1 = t, k1 = 0;      // step s+1
	bool live1 = live;
	issue(0, t, 0, live);
	advance(t1, k1, live1);
	if (!seg_start(k1)) { // same task (and segment), next k-offset: same rows
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			rows[1][r] = rows[0][r];
		}
	}
	issue(1, t1, k1, live1);
	stage();
	f32x4 xr[XR > 0 ? XR : 1][XRegs<DB>::NF];
	if constexpr (XR > 0) {
#pragma unroll
		for (int c_ = 0; c_ < XR; ++c_) {
#pragma unroll
			for (int i = 0; i < XRegs<DB>::NF; ++i) {
				xr[c_][i] = ((const f32x4*)xs4)[c_ * Fmt<DB>::CS + i * Fmt<DB>::ROW + lane];
			}
		}
	}
#ifdef CALM_TIMELINE
	tl[1] = wall_clock64();
#endif
	if (!live) {
#ifdef CALM_TIMELINE
		tl_flush();
#endif
		return;
	}
#pragma unroll
	for (int r = 0; r < NR; ++r) {
		acc2[r] = (f32x2){0.f, 0.f};
	}
	for (;;) {
#pragma unroll
		for (int ph = 0; ph < 2; ++ph) {
			// (t, k0) lives in tile[ph]; (t1, k1) in tile[ph ^ 1]
			const int seg = SEGS ? k0 / cps : 0, kl = SEGS ? k0 - seg * cps : k0; // segment, chunk offset inside it
			const bool last_k = SEGS ? kl + U >= cps : (k0 + U) * 64 >= nl;          // the row's (segment's) last step
			if (last_k && seg == 0) {
				aux_of(t, aux);
			}
			if constexpr (XR > 0) {
				tile_fma_regs<DB, NR, U, XR>(tile[ph], acc2, xr, XR == U ? 0 : ph * U);
			} else {
				tile_fma<DB, NR, U, FULL>(tile[ph], acc2, SEGS ? xs4 + seg * cpi * Fmt<DB>::CS : xs4, kl, nl, lane);
			}
#ifdef CALM_TIMELINE
			if (tl[2] == 0) {
				asm volatile("" ::"v"(acc2[0][0]));
				tl[2] = wall_clock64();
			}
#endif
			int t2 = t1, k2 = k1;
			bool live2 = live1;
			advance(t2, k2, live2);
			if (!seg_start(k2) && live2) { // continues the row of step s+1: same rows
#pragma unroll
				for (int r = 0; r < NR; ++r) {
					rows[ph][r] = rows[ph ^ 1][r];
				}
			}
			issue(ph, t2, k2, live2);
			if (last_k) {
#pragma unroll
				for (int r = 0; r < NR; ++r) {
					acc[r] = wave_sum63(acc2[r][0] + acc2[r][1]); // valid in lane RED_LANE
					acc2[r] = (f32x2){0.f, 0.f};
				}
				if constexpr (SEGS) {
					epi(t, seg, acc, aux);
				} else {
					epi(t, acc, aux);
				}
			}
			if (!live1) {
#ifdef CALM_TIMELINE
				tl_flush();
#endif
				return;
			}
			t = t1, k0 = k1;
			t1 = t2, k1 = k2, live1 = live2;
		}
	}
}

// FULL (rows are whole KiB chunks) is a KERNEL template parameter picked by the host: carrying both variants
// in one kernel doubled its code size for a branch that never changes (kernels this short feel their
// instruction-cache warm-up).
template <int DB, int NR, int U, bool FULL, int XR = 0, class RowsFn, class PreFn, class StageFn, class AuxFn, class EpiFn>
__device__ __forceinline__ void run_rows(int ntasks, int first, int stride, int n, const float4* xs4, const void* dummy, RowsFn rows_of, PreFn pre,
                                         StageFn stage, AuxFn aux_of, EpiFn epi) {
	run_rows_impl<DB, NR, U, FULL, false, XR>(ntasks, first, stride, n, 1, xs4, dummy, rows_of, pre, stage, aux_of, epi);
}
// ... over rows of `segs` segments of n columns each (run_rows_impl: SEGS)
template <int DB, int NR, int U, bool FULL, class RowsFn, class PreFn, class StageFn, class AuxFn, class EpiFn>
__device__ __forceinline__ void run_rows_segs(int ntasks, int first, int stride, int n, int segs, const float4* xs4, const void* dummy, RowsFn rows_of, PreFn pre,
                                              StageFn stage, AuxFn aux_of, EpiFn epi) {
	run_rows_impl<DB, NR, U, FULL, true, 0>(ntasks, first, stride, n, segs, xs4, dummy, rows_of, pre, stage, aux_of, epi);
}

// Workgroup shape of the matvec kernels that stage a dim-sized vector (k_qkv, k_attn_out, k_ffn_up, k_output): 256 threads, two
// workgroups per CU.  (One 512-thread workgroup per CU: - 4 %; a barrier, or a wait for the vector to have landed, between the
// staging loads and the first tile loads: no change -- profiles/HISTORY.md section 5c.)
constexpr int WG_THREADS = 256, WG_WAVES = WG_THREADS / 64;
__device__ __forceinline__ void stage_first_barrier() {
}

// Tile shape per kernel and weight format: NR rows per task x U 1-KiB chunks of each per step, two tiles in flight per wave.
// Shallower tiles start multiplying sooner, put fewer weight bytes ahead of the activation vector in the start-up burst (every wave
// asks for its first two tiles before it builds the LDS image: with 8-KiB tiles that is 32 MB chip-wide, which takes 4-5 us to
// deliver and the vector arrives inside it) and leave no half-empty last step on rows of 4 n + 2 chunks; deeper ones keep more
// bytes in flight, more rows share one read of the image.
//   fp8 / fp16 (2 rows; Mistral-7B fp8 shape, profiles/r03_startup_experiments.txt): k_qkv is best at 4 chunks (7.7 us; 8.1-8.4 at
//     2; 10.0 at 1), k_attn_out at 1 or 2 (5.2-5.3 against 5.9; at 1 DBRX's 6-chunk rows lose: 11.9 against 9.9 us), k_ffn_up and
//     k_output at 2 (20.3 / 21.8 against 21.1 / 22.3); k_ffn_down: 2 on rows of 4 n + 2 chunks (infer_hip.hip), else 4.
//   gf4 (Llama-3-8B shape, profiles/r03_gf4_tables.txt; rows x chunks -> us): its kernels move half the bytes, so the start-up
//     burst weighs twice as much.  k_qkv 4x2 8.6, 4x1 7.9, 2x2 6.8, 2x1 7.3-8.2; k_attn_out 4.9, 4.7, 4.5, 4.45; k_ffn_up 15.5, 13.9,
//     15.8, 14.9-16.5; k_ffn_down (7-chunk rows) 2x7 12.8, 4x2 14.2, 4x1 13.5, 2x2 9.7, 2x1 10.0; k_output (up to 4 workgroups per CU)
//     48.9, 45.2, 44.2, 47.2.
//     Round 4: with the activations in registers (XREG) no row shares an image read any more and 2 x 2 -- 14 rounds of tasks, so the skewed
//     deal applies -- is k_ffn_up's best: 4x1 13.70, 2x1 14.15, 2x2 13.43 us (profiles/r04_gf4.txt).
// Task ranges of a wave when the grid is two workgroups per CU and the FIRST-dispatched one is given more tasks (`cut` > 0:
// SKEW_PERCENT in infer_hip.hip; "forms" 1: even deal): a CU's older workgroup wins its memory queue and used to leave 2-3 us before the younger one, whose waves then ran the
// kernel's tail alone at half the bytes in flight (profiles/r02_kernel_timeline.txt).  Blocks [0, G/2) deal tasks [0, cut) among
// their waves, blocks [G/2, G) tasks [cut, ntasks).  Placement-independent: another dispatch order only changes who finishes when.
struct TaskRange {
	int first, stride, ntasks;
};
__device__ __forceinline__ TaskRange task_range(int ntasks, int waves_per_block, int wave, int cut) {
	const int b = blockIdx.x, g = gridDim.x;
	if (cut <= 0 || (g & 1)) {
		return {b * waves_per_block + wave, g * waves_per_block, ntasks};
	}
	const int half = g >> 1, stride = half * waves_per_block;
	if (b < half) {
		return {b * waves_per_block + wave, stride, cut};
	}
	return {cut + (b - half) * waves_per_block + wave, stride, ntasks};
}

enum KernelId { KS_QKV, KS_ATTN_OUT, KS_FFN_UP, KS_FFN_DOWN, KS_OUTPUT };
// XREG kernels ("forms" 1 turns them off): the launcher picks them where the input vector has exactly 4096 columns at fp8 (rows of 4 chunks) or at
// gf4 (2 chunks), 2048 at fp16 (4 chunks) -- the BASELINE models' dim -- and every lane keeps its float4s of the image in registers
// (run_rows_impl XR: 64-72 VGPRs at fp8 / gf4, 32 at fp16)
template <int DB, bool XREG>
constexpr int xreg_chunks() {
	return XREG ? (DB == 4 ? 2 : 4) : 0;
}
template <int NR_, int U_>
struct ShapeOf {
	static constexpr int NR = NR_, U = U_;
};
template <int K, class Q, class A, class F, class D, class O>
using ShapePick = std::conditional_t<K == KS_QKV, Q, std::conditional_t<K == KS_ATTN_OUT, A, std::conditional_t<K == KS_FFN_UP, F, std::conditional_t<K == KS_FFN_DOWN, D, O>>>>;
template <int DB, int K>
struct KShape {
	//                                                            k_qkv          k_attn_out     k_ffn_up       k_ffn_down     k_output
	using S = std::conditional_t<DB == 4, ShapePick<K, ShapeOf<2, 2>, ShapeOf<2, 1>, ShapeOf<2, 2>, ShapeOf<2, 2>, ShapeOf<2, 2>>,    // gf4
	                             ShapePick<K, ShapeOf<2, 4>, ShapeOf<2, 2>, ShapeOf<2, 2>, ShapeOf<2, 4>, ShapeOf<2, 2>>>;   // fp8, fp16
	static constexpr int NR = S::NR, U = S::U;
	// resident 256-thread workgroups per CU the kernel's grid is sized for (0: the common default, 2): the gf4 classifier takes 4
	static constexpr int BPC = (K == KS_OUTPUT && DB == 4) ? 4 : 0;
};

__device__ __forceinline__ float clipf(float x, float v) {
	return x < -v ? -v : (x > v ? v : x); // src/infer.c:307-309
}

// ---------------------------------------------------------------- kernels ---------------------

// token state + embedding row + this position's RoPE table.   (src/infer.c:329-347)
// tok_src != nullptr: the token is read from device memory (device-side greedy decode).
template <int DB>
__global__ void k_begin_token(TokState* ts, int token, const int* tok_src, int pos, int kv_sink, int kv_pos, int kv_len, float* x, const void* embed,
                              int dim, const float* rope_freq, float2* rope_cs, int half_hd) {
	if (tok_src) {
		token = *tok_src;
	}
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) {
		ts->token = token;
		ts->pos = pos;
		ts->kv_sink = kv_sink;
		ts->kv_pos = kv_pos;
		ts->kv_len = kv_len;
		ts->epoch = ts->epoch + 1;
	}
	if (embed && i < dim) { // embed == nullptr: a later pipeline stage; x already holds the residual stream
		x[i] = decode_elem<DB>(embed, (size_t)token * dim + i);
	}
	if (i < half_hd) {
		float val = (float)pos * rope_freq[i]; // src/infer.c:227-229
		rope_cs[i] = make_float2(cosf(val), sinf(val));
	}
}

// Advance the two attention-sink keys of every layer by one RoPE position (src/infer.c:383-394).
// K cache layout: [layer][kv_head][seq_len][head_dim].  grid = (ceil(sink pairs / block), n_layers)
template <int KVB>
__global__ void k_rotate_sink(void* kc, const float2* rope_cs1, int n_kv_heads, int head_dim, int seq_len, int kv_sink) {
	int half_hd = head_dim >> 1;
	int idx = blockIdx.x * blockDim.x + threadIdx.x; // over (kv_head, r, pair)
	int total = n_kv_heads * kv_sink * half_hd;
	if (idx >= total) {
		return;
	}
	int pair = idx % half_hd;
	int r = (idx / half_hd) % kv_sink;
	int h = idx / (half_hd * kv_sink);
	size_t off = (((size_t)blockIdx.y * n_kv_heads + h) * seq_len + r) * head_dim + 2 * pair;
	float2 cs = rope_cs1[pair];
	if constexpr (KVB == 16) {
		__half2* p = (__half2*)((__half*)kc + off);
		float2 v = __half22float2(*p);
		*p = __floats2half2_rn(v.x * cs.x - v.y * cs.y, v.x * cs.y + v.y * cs.x);
	} else {
		unsigned short* p = (unsigned short*)((unsigned char*)kc + off);
		unsigned short b = *p;
		f32x2 v = bf8x2_lo(b);
		float a = v[0] * cs.x - v[1] * cs.y, c = v[0] * cs.y + v[1] * cs.x;
		*p = e5m2x2_sat(a, c);
	}
}

// Rows [r0, r1) of every layer's value cache copied into the transposed cache (attn_vt_offset).  The decode step writes V^T only in
// steps whose attention is split (the only reader, k_attn_vt): a sequence's first split step -- and a prompt chunk behind rows that
// came from unsplit steps -- first brings the transposed copy up to date with this (infer_hip.hip run_step, Ctx::vt_rows).  Rare
// (once per sequence): one thread per element.  grid = (ceil((r1 - r0) * kv_dim / 256), n_layers)
template <int KVB>
__global__ void k_vt_backfill(const void* vc, void* vt, size_t layer_elems, int kv_dim, int head_dim, int seq_len, int r0, int r1) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (r1 - r0) * kv_dim) {
		return;
	}
	const int p = r0 + idx / kv_dim, row = idx % kv_dim; // row = (kv head, dim)
	const size_t src = (size_t)blockIdx.y * layer_elems + ((size_t)(row / head_dim) * seq_len + p) * head_dim + row % head_dim;
	const size_t dst = (size_t)blockIdx.y * layer_elems + attn_vt_offset(row, p, head_dim, seq_len, KVB / 8);
	if constexpr (KVB == 16) {
		((unsigned short*)vt)[dst] = ((const unsigned short*)vc)[src];
	} else {
		((unsigned char*)vt)[dst] = ((const unsigned char*)vc)[src];
	}
}

struct QkvArgs {
	const float* x;
	const float* norm_w;
	const void *wq, *wk, *wv;
	const float* bqkv;
	float* q;
	void *kc, *vc; // this layer's K / V cache: [kv_head][seq_len][head_dim]
	void* vt;      // this layer's transposed V cache [kv_head][head_dim][seq_len] (k_attn_vt), or nullptr
	float* xb_dump;
	const TokState* ts;
	const float2* rope_cs;
	int dim, q_dim, kv_dim, head_dim, seq_len;
	float eps, clip;
	int ln;
};

// attention norm + fused q/k/v matvec + bias + clip + RoPE + KV append   (src/infer.c:352-381)
// task = NR consecutive rows of the concatenated [wq; wk; wv]; rows come in RoPE pairs (2i, 2i+1).
// The arguments a wave needs before it can ask for its first byte come FIRST and as scalars: with kernel-argument preloading
// (hipcc -mllvm -amdgpu-kernarg-preload-count, calm_amd/build.py) the dispatcher delivers the leading dwords in SGPRs with the
// wave instead of the wave fetching them from memory -- a round trip at the head of every launch.  The struct carries the rest
// (and copies of the leading ones, which the kernel does not read; the struct itself is never written: a modified by-value
// struct argument is copied to scratch memory, +4 us per launch when that was tried).
// HALF: tiles half as deep -- for matrices so small that a wave's share is less than one full tile (TinyLlama's 10.5 MB at fp16: 6.3 ->
// 5.45 us; the host decides, launch_qkv)
// In-launch hand-off words of k_qkv_attn (below): this token's q and its k / v rows AS CACHED, one 8-byte {value, tag} granule per
// element, indexed like the rows of [wq; wk; wv].  One buffer serves every layer: the tag names (decode step, layer).
struct FuseArgs {
	unsigned long long* gran;
	float* out;      // (q_dim) attention output
	unsigned* err;   // raised when a wait ran into its bound (never, unless a producer died)
	int n_heads, kv_mul;
	unsigned layer;  // < 256
	unsigned salt;   // added to the step count: 0 in a decode step; perf_stage_hip, which repeats one launch without a begin-token kernel, counts it up
};
__device__ __forceinline__ unsigned fuse_tag(const TokState* ts, const FuseArgs& f) {
	return ((ts->epoch + f.salt) << 8) | f.layer;
}
// written by ONE relaxed agent-scope (write-through, sc1) 8-byte store: the data is its own flag -- no drain of the writing wave's
// memory queue (its next weight tiles are in flight), no counter, no fence
__device__ __forceinline__ void gran_store(unsigned long long* g, float v, unsigned tag) {
	__hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_add_epoch(TokState* ts, unsigned n) {
	ts->epoch += n;
}

template <int DB, int KVB, int V, bool FULL, bool HALF, bool XREG, bool FUSED>
__device__ __forceinline__ void qkv_rows(const float* x, const float* norm_w, int dim, int q_dim, int kv_dim, const QkvArgs& a, const FuseArgs* f, int bid, int nblocks,
                                         unsigned char* smem) {
	constexpr int NR = KShape<DB, KS_QKV>::NR, U = (HALF && KShape<DB, KS_QKV>::U > 1) ? KShape<DB, KS_QKV>::U / 2 : KShape<DB, KS_QKV>::U;
	float4* xs4 = (float4*)smem;
	float* red = (float*)(xs4 + xs_slots<DB>(dim));
	const int rows_total = q_dim + 2 * kv_dim;
	const int ntasks = rows_total / NR;
	const size_t row_bytes = (size_t)dim * DB / 8;
	const int lane = lane_id();

	auto row_ptr = [&](int j) -> const unsigned char* {
		// j is wave-uniform: keep the three-way choice in scalar selects (an if-chain over the three
		// kernel-argument pointers gets turned into a scratch-memory lookup table by the optimiser)
		j = __builtin_amdgcn_readfirstlane(j);
		const bool is_q = j < q_dim, is_k = j < q_dim + kv_dim;
		const unsigned char* base = (const unsigned char*)(is_q ? a.wq : (is_k ? a.wk : a.wv)); // (from the struct: as leading scalars the select became a scratch table)
		const int jl = j - (is_q ? 0 : (is_k ? q_dim : q_dim + kv_dim));
		return base + (size_t)jl * row_bytes;
	};
	auto rows_of = [&](int t, const unsigned char*(&rows)[NR]) {
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			rows[r] = row_ptr(t * NR + r);
		}
	};
	StageRegs<V, true> sr;
	auto pre = [&]() { stage_load<WG_THREADS>(sr, x, norm_w); stage_first_barrier(); };
	float nscale = 1.f; // what the norm leaves to the epilogue (stage_finish)
	auto stage = [&]() { nscale = stage_finish<DB, WG_THREADS>(sr, xs4, red, x, norm_w, dim, a.eps, a.ln != 0, a.xb_dump, (int)blockIdx.x - bid); };
	const int kv_pos = a.ts->kv_pos; // scalar load issued at kernel start, long before any epilogue
	unsigned tag = 0;
	if constexpr (FUSED) {
		tag = fuse_tag(a.ts, *f);
	}
	// aux = (cos, sin) of each row pair's RoPE angle, fetched before the task's last multiply-add
	auto aux_of = [&](int t, float(&aux)[NR]) {
#pragma unroll
		for (int r = 0; r < NR; r += 2) {
			int j = t * NR + r;
			int jl = j < q_dim ? j : j - q_dim;
			float2 cs = a.rope_cs[(jl % a.head_dim) >> 1]; // v rows: an unused but in-bounds entry
			aux[r] = cs.x;
			aux[r + 1] = cs.y;
		}
	};
	auto epi = [&](int t, float(&acc)[NR], float(&aux)[NR]) {
		if (lane != RED_LANE) {
			return;
		}
#pragma unroll
		for (int r = 0; r < NR; r += 2) {
			int j = t * NR + r; // even row of a pair
			float v0 = acc[r] * nscale, v1 = acc[r + 1] * nscale;
			if (a.bqkv) {
				v0 += a.bqkv[j];
				v1 += a.bqkv[j + 1];
			}
			v0 = clipf(v0, a.clip);
			v1 = clipf(v1, a.clip);
			if (j < q_dim + kv_dim) { // q or k: rotate the pair (src/infer.c:223-236)
				float r0 = v0 * aux[r] - v1 * aux[r + 1];
				float r1 = v0 * aux[r + 1] + v1 * aux[r];
				v0 = r0;
				v1 = r1;
			}
			if (j < q_dim) {
				if constexpr (FUSED) { // the attention workgroups of this launch are the only readers
					gran_store(f->gran + j, v0, tag);
					gran_store(f->gran + j + 1, v1, tag);
				} else {
					*(float2*)(a.q + j) = make_float2(v0, v1);
				}
			} else {
				int jl = j - q_dim;
				void* cache = a.kc;
				if (jl >= kv_dim) {
					jl -= kv_dim;
					cache = a.vc;
				}
				size_t off = ((size_t)(jl / a.head_dim) * a.seq_len + kv_pos) * a.head_dim + (jl % a.head_dim);
				const bool tr = cache == a.vc && a.vt;
				const size_t offt = tr ? attn_vt_offset(jl, kv_pos, a.head_dim, a.seq_len, KVB / 8) : 0; // the same element in the transposed cache: row (kv head, dim) = jl
				constexpr int next_d = VT_BLOCK_BYTES / (KVB / 8);                                          // ... and the next dim of the same position
				if constexpr (KVB == 16) {
					const __half2 h = __floats2half2_rn(v0, v1); // RNE, as (half)x: src/infer.c:378-381
					*(__half2*)((__half*)cache + off) = h;
					if (tr) {
						((__half*)a.vt)[offt] = __low2half(h);
						((__half*)a.vt)[offt + next_d] = __high2half(h);
					}
					if constexpr (FUSED) { // this step's attention reads the row as later steps will find it in the cache
						gran_store(f->gran + j, __low2float(h), tag);
						gran_store(f->gran + j + 1, __high2float(h), tag);
					}
				} else {
					const unsigned short b2 = e5m2x2_sat(v0, v1);
					*(unsigned short*)((unsigned char*)cache + off) = b2;
					if (tr) {
						((unsigned char*)a.vt)[offt] = (unsigned char)(b2 & 0xff);
						((unsigned char*)a.vt)[offt + next_d] = (unsigned char)(b2 >> 8);
					}
					if constexpr (FUSED) {
						gran_store(f->gran + j, bf8_byte0(b2 & 0xffu), tag);
						gran_store(f->gran + j + 1, bf8_byte0((unsigned)b2 >> 8), tag);
					}
				}
			}
		}
	};
	// k_qkv deals tasks wave by wave (wave w of block b starts at task 4 b + w); the fused launch, whose ncu - n_heads workgroups leave the
	// last round of tasks partly empty (3072 tasks over 896 waves: 3.4 each), deals them block by block -- task t to block t % nblocks --
	// so that every CU gets the same 13 or 14 of them instead of 16 on some and 12 on others
	const int first = FUSED ? wave_id() * nblocks + bid : bid * WG_WAVES + wave_id();
	run_rows<DB, NR, U, FULL, xreg_chunks<DB, XREG>()>(ntasks, first, nblocks * WG_WAVES, dim, xs4, x, rows_of, pre, stage, aux_of, epi);
}

template <int DB, int KVB, int V, bool FULL, bool HALF, bool XREG = false>
__global__ __launch_bounds__(WG_THREADS) void k_qkv(const float* x, const float* norm_w, int dim, int q_dim, int kv_dim, QkvArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	qkv_rows<DB, KVB, V, FULL, HALF, XREG, false>(x, norm_w, dim, q_dim, kv_dim, a, nullptr, (int)blockIdx.x, (int)gridDim.x, smem);
}

// ---- attention --------------------------------------------------------------------------------

struct AttnArgs {
	const float* q;
	const void *kc, *vc; // this layer's caches
	float* out;          // (q_dim) normalised attention output, written when n_split == 1
	float* partial;      // (n_heads, n_split, pstride) when n_split > 1: o[head_dim], m, l (k_attn_gqa: pstride = head_dim + 2; k_attn_vt: head_dim + 4)
	const TokState* ts;
	int head_dim, kv_mul, seq_len, n_split;
	// batched prompt ingestion (prefill.hip.h: k_pf_attn): token b of the chunk reads q + b * pf_stride, attends to
	// cache rows [0, pf_kv0 + b] and writes row b of the fragment-major matrix `out`; pf_nb tokens in the chunk
	int pf_kv0, pf_stride, pf_nb;
};

// merge two online-softmax states (m, l, o[8])
__device__ __forceinline__ void sm_merge(float& m, float& l, float (&o)[8], float m2, float l2, const float (&o2)[8]) {
	float M = fmaxf(m, m2);
	float c1 = (m == -INFINITY) ? 0.f : __expf(m - M);
	float c2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - M);
	l = l * c1 + l2 * c2;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		o[i] = o[i] * c1 + o2[i] * c2;
	}
	m = M;
}

constexpr int ATTN_ROUND_TILES = 64; // wave-tiles of 64/LPR positions per round of the workgroup: NW waves x 64 / NW tiles each

// Short-context attention: one workgroup (16 waves) per query head, the whole cached range in one or two rounds, no
// merge pass (longer contexts: k_attn_gqa + k_attn_merge).  LPR lanes cover one cached row (8 dims per lane, one
// 16-byte load for fp16), so a wave-load covers 64/LPR positions; the 16 waves interleave tiles of positions.
// Scores, max-subtracted softmax and the V mix (src/infer.c:238-267) are computed in one pass with running
// (max, sum, out) per lane group -- algebraically the same result as the reference's three loops.
// The kernel is latency, not bandwidth: its K/V rows were last touched a token ago and come from HBM.  Rounds are
// loaded one ahead of the arithmetic.  (Issuing the first round before kv_len is known -- clamped to the cache instead
// of the live range -- measured the same 4.7 us and fetched 4 MB per launch of rows nobody needs: not kept.)
// NW: waves per workgroup (16 / 8 / 4), each holding 64 / NW tiles in flight -- the same 256 positions (at head size 128) per
// round either way; fewer, fatter waves start and merge faster, more waves hide more latency (picked by measurement, infer_hip.hip).
template <int KVB, int LPR, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn(const TokState* ts, const float* qin, const void* kc, const void* vc, int head_dim, int kv_mul, int seq_len, AttnArgs a) {
	constexpr int RPW = 64 / LPR; // positions per wave-load
	constexpr int ATTN_BLOCK = NW * 64;
	constexpr int UA = ATTN_ROUND_TILES / NW; // tiles in flight per wave
	constexpr int STEP = NW * RPW * UA; // positions per round of the workgroup
	__shared__ float sm_m[NW], sm_l[NW];
	__shared__ float sm_o[NW][LPR * 8];

	const int lane = lane_id(), wave = wave_id();
	const int h = blockIdx.x;
	const int kvh = h / kv_mul;
	const int r = lane % LPR, g = lane / LPR;
	const bool dvalid = r * 8 < head_dim;
	const int d0 = dvalid ? r * 8 : 0; // lanes past head_dim (non power-of-two heads) shadow dims 0..7 and are masked

	constexpr int EB = KVB / 8; // bytes per element
	using Raw = std::conditional_t<KVB == 16, u32x4, u32x2>; // 8 cached elements
	const unsigned char* kbase = (const unsigned char*)kc + ((size_t)kvh * seq_len * head_dim + d0) * EB;
	const unsigned char* vbase = (const unsigned char*)vc + ((size_t)kvh * seq_len * head_dim + d0) * EB;
	const size_t rstride = (size_t)head_dim * EB;
	Raw kw[UA], vw[UA];
	auto load_round = [&](int tb, int last_row) { // always issued, clamped into [0, last_row]
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			const int t = min(tb + u * NW * RPW + g, last_row);
			kw[u] = *(const Raw*)(kbase + (size_t)t * rstride);
			vw[u] = *(const Raw*)(vbase + (size_t)t * rstride);
		}
	};
#ifdef CALM_TIMELINE
	unsigned long long tl[6];
	tl[0] = wall_clock64();
#endif
	const int kv_len = ts->kv_len;
#ifdef CALM_TIMELINE
	asm volatile("" ::"s"(kv_len));
	tl[1] = wall_clock64(); // the cache length is known
#endif
	load_round(wave * RPW, kv_len - 1);

	float qv[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		float qi = qin[h * head_dim + d0 + i];
		qv[i] = dvalid ? qi : 0.f;
	}
	const float sqrt_hd = sqrtf((float)head_dim);

	float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		o[i] = 0.f;
	}

	for (int tb = wave * RPW; tb < kv_len; tb += STEP) {
		float kf[UA][8], vf[UA][8];
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			if constexpr (KVB == 16) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					kf[u][2 * i] = half_bits_to_float((unsigned short)(kw[u][i] & 0xffff));
					kf[u][2 * i + 1] = half_bits_to_float((unsigned short)(kw[u][i] >> 16));
					vf[u][2 * i] = half_bits_to_float((unsigned short)(vw[u][i] & 0xffff));
					vf[u][2 * i + 1] = half_bits_to_float((unsigned short)(vw[u][i] >> 16));
				}
			} else {
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					f32x2 k0 = bf8x2_lo(kw[u][i]), k1 = bf8x2_hi(kw[u][i]);
					f32x2 v0 = bf8x2_lo(vw[u][i]), v1 = bf8x2_hi(vw[u][i]);
					kf[u][4 * i] = k0[0], kf[u][4 * i + 1] = k0[1], kf[u][4 * i + 2] = k1[0], kf[u][4 * i + 3] = k1[1];
					vf[u][4 * i] = v0[0], vf[u][4 * i + 1] = v0[1], vf[u][4 * i + 2] = v1[0], vf[u][4 * i + 3] = v1[1];
				}
			}
		}
		float s[UA];
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			float d = 0.f;
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				d = fmaf(qv[i], kf[u][i], d);
			}
			s[u] = group_sum<LPR>(d);
		}
		if (tb + STEP < kv_len) { // wave-uniform: one round ahead
			load_round(tb + STEP, kv_len - 1);
		}
		bool valid[UA];
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			valid[u] = tb + u * NW * RPW + g < kv_len;
			s[u] = valid[u] ? s[u] / sqrt_hd : -INFINITY; // src/infer.c:247
		}
		float mn = m;
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			mn = fmaxf(mn, s[u]);
		}
		if (mn != -INFINITY) {
			float c = (m == -INFINITY) ? 0.f : __expf(m - mn);
			l *= c;
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				o[i] *= c;
			}
#pragma unroll
			for (int u = 0; u < UA; ++u) {
				float p = valid[u] ? __expf(s[u] - mn) : 0.f;
				l += p;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					o[i] = fmaf(p, vf[u][i], o[i]);
				}
			}
			m = mn;
		}
	}

#ifdef CALM_TIMELINE
	asm volatile("" ::"v"(o[0]), "v"(l));
	tl[2] = wall_clock64(); // the wave's positions are folded in
#endif
	// merge the RPW lane groups of the wave
#pragma unroll
	for (int ofs = LPR; ofs < 64; ofs <<= 1) {
		float m2 = __shfl_xor(m, ofs), l2 = __shfl_xor(l, ofs), o2[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			o2[i] = __shfl_xor(o[i], ofs);
		}
		sm_merge(m, l, o, m2, l2, o2);
	}
	// merge the waves through LDS
	if (g == 0) {
		if (r == 0) {
			sm_m[wave] = m;
			sm_l[wave] = l;
		}
		if (dvalid) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				sm_o[wave][d0 + i] = o[i];
			}
		}
	}
#ifdef CALM_TIMELINE
	asm volatile("" ::"v"(o[0]), "v"(l));
	tl[3] = wall_clock64(); // lane groups merged, partials in LDS
#endif
	__syncthreads();
#ifdef CALM_TIMELINE
	tl[4] = wall_clock64(); // every wave is here
#endif
	// Merge the NW wave partials with one THREAD per output dim: the weights exp(m_w - M) are recomputed by every
	// thread (NW exps), then one pass over the partials -- all of it parallel over head_dim threads.  (One lane group
	// folding the waves in one after the other was a chain of NW dependent exp + rescale steps: ~1 us of this kernel.)
	for (int d = threadIdx.x; d < head_dim; d += ATTN_BLOCK) {
		float M = sm_m[0];
#pragma unroll
		for (int w = 1; w < NW; ++w) {
			M = fmaxf(M, sm_m[w]);
		}
		float L = 0.f, O = 0.f;
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const float e = (sm_m[w] == -INFINITY) ? 0.f : __expf(sm_m[w] - M); // a wave without positions
			L = fmaf(sm_l[w], e, L);
			O = fmaf(sm_o[w][d], e, O);
		}
		a.out[h * head_dim + d] = O / L;
	}
#ifdef CALM_TIMELINE
	{
		tl[5] = wall_clock64();
		const unsigned w = blockIdx.x * NW + wave;
		if (lane == 0 && calm_tl_buf && w < calm_tl_waves) {
			unsigned long long* ob = calm_tl_buf + (size_t)w * 8;
			ob[0] = tl[0], ob[1] = tl[1], ob[2] = tl[2], ob[3] = tl[5], ob[4] = tl[3], ob[5] = tl[4];
		}
	}
#endif
}

// ---- k_qkv_attn: short-context attention INSIDE k_qkv's launch ---------------------------------
//
// The reference runs the whole layer in one cooperative kernel (src/infer.cu:441-556); here a kernel boundary is cheaper than a grid
// barrier, but the boundary between k_qkv and k_attn bought nothing except its cost: k_attn's K / V rows were
// written by EARLIER tokens, and the one round trip to memory they cost (~1.9 us behind a boundary whose L2s come up empty,
// profiles/r05_attn_floor.txt) could have been under way since k_qkv started.  So: the first n_heads workgroups of this launch take a
// query head each, ask for every cached row of their kv head at once (registers: 16 wave-tiles of K and of V per wave), and only
// then wait -- for the 128 q values of THEIR head, never for the grid: the row engine's epilogue publishes every q / k / v value as an
// 8-byte {value, tag} granule by one write-through store (the data is its own flag: the producing wave drains nothing -- its next
// weight tiles are in flight -- and counts nothing), the attention wave re-reads its head's granules past the caches until every tag
// names this (decode step, layer).  q rows come first in the task order and this token's k / v rows last, so a head has folded the
// old positions in long before the launch ends and the tail behind the last producer is ONE position: a dot product, an exp, 128
// multiply-adds.  Producers never wait for anything, consumers only for producers: nothing can hang whatever the dispatch order
// (the wait is bounded all the same).  Matches src/infer.c:238-267,397-406 like k_attn.
constexpr int FUSE_UA = 64 / WG_WAVES; // wave-tiles of 64 / LPR cached rows a wave holds of K and of V (16 at 4 waves: 128 VGPRs at binary16)
__host__ __device__ constexpr int fuse_max_kv(int lpr) {
	return WG_WAVES * (64 / lpr) * FUSE_UA; // cached rows (the new one included) one workgroup covers: 256 at head size 128, 512 at 64
}

// 16 bytes past the L1 (agent scope), waited for in the same statement (the compiler does not count it)
__device__ __forceinline__ u32x4 load16_agent(const void* p) {
	u32x4 v;
	asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
	return v;
}
constexpr unsigned long long FUSE_WAIT_TICKS = 5000000ull; // 50 ms of the 100 MHz wall clock

template <int KVB>
using KvRaw = std::conditional_t<KVB == 16, u32x4, u32x2>; // 8 cached elements
template <int KVB>
__device__ __forceinline__ float dot8_raw(const KvRaw<KVB>& k, const float (&q)[8]) {
	float d = 0.f;
	if constexpr (KVB == 16) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			d = fma_mix_lo(k[i], q[2 * i], d);
			d = fma_mix_hi(k[i], q[2 * i + 1], d);
		}
	} else {
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const f32x2 lo = bf8x2_lo(k[i]), hi = bf8x2_hi(k[i]);
			d = fmaf(q[4 * i], lo[0], d), d = fmaf(q[4 * i + 1], lo[1], d), d = fmaf(q[4 * i + 2], hi[0], d), d = fmaf(q[4 * i + 3], hi[1], d);
		}
	}
	return d;
}
// exp2 for a value that asm statements go on to read.  gfx940+ wants one wait state between a transcendental instruction and a
// dependent ordinary VALU instruction; hipcc's hazard pass inserts it between ITS instructions and does not look inside an asm string
// (round 6: NaNs whenever the scheduler put an asm v_fma_mix_f32 right behind the v_exp_f32 that made its factor).  The copy below is
// the one reader of the exponential's register and carries the wait state itself.
__device__ __forceinline__ float exp2_for_asm(float x) {
	const float e = __builtin_amdgcn_exp2f(x);
	float r;
	asm("s_nop 0\n\tv_mov_b32 %0, %1" : "=v"(r) : "v"(e));
	return r;
}
template <int KVB>
__device__ __forceinline__ void axpy8_raw(float p, const KvRaw<KVB>& v, float (&o)[8]) {
	if constexpr (KVB == 16) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			o[2 * i] = fma_mix_lo(v[i], p, o[2 * i]);
			o[2 * i + 1] = fma_mix_hi(v[i], p, o[2 * i + 1]);
		}
	} else {
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const f32x2 lo = bf8x2_lo(v[i]), hi = bf8x2_hi(v[i]);
			o[4 * i] = fmaf(p, lo[0], o[4 * i]), o[4 * i + 1] = fmaf(p, lo[1], o[4 * i + 1]), o[4 * i + 2] = fmaf(p, hi[0], o[4 * i + 2]), o[4 * i + 3] = fmaf(p, hi[1], o[4 * i + 3]);
		}
	}
}

// LDS the attention role needs (it shares the launch's dynamic allocation with the row engine's image)
__host__ __device__ constexpr size_t fuse_lds_bytes(int lpr) {
	return (size_t)(WG_WAVES + WG_WAVES * (64 / lpr) + WG_WAVES * 64 * 8 + WG_WAVES * lpr * 8) * sizeof(float);
}

template <int KVB, int LPR>
__device__ __forceinline__ void attn_fused_role(const QkvArgs& a, const FuseArgs& f, int q_dim, int kv_dim, unsigned char* smem) {
	constexpr int RPW = 64 / LPR, NW = WG_WAVES, UA = FUSE_UA;
	constexpr int EB = KVB / 8;
	using Raw = KvRaw<KVB>;
	float* sm_m = (float*)smem;         // [NW]            every wave's maximum (base-2 scores)
	float* sm_l = sm_m + NW;            // [NW][RPW]       ... and each of its lane groups' sum of weights
	float* sm_o = sm_l + NW * RPW;      // [NW][RPW][LPR * 8]  ... and weighted value sum
	float* sm_q = sm_o + NW * 64 * 8;   // [NW][LPR * 8]: every wave's copy of q

	const int lane = lane_id(), wave = wave_id();
	const int head_dim = a.head_dim;
	// workgroup b sits on XCD b % 8: the query heads of one kv head on the same one where the head counts allow (their rows then come
	// from memory once per XCD); any placement is correct
	const int n_kv = f.n_heads / f.kv_mul;
	const int b = blockIdx.x;
	const int kvh = b % n_kv, h = kvh * f.kv_mul + b / n_kv;
	const int r = lane % LPR, g = lane / LPR;
	const bool dvalid = r * 8 < head_dim;
	const int d0 = dvalid ? r * 8 : 0;
	const unsigned char* kbase = (const unsigned char*)a.kc + ((size_t)kvh * a.seq_len * head_dim + d0) * EB;
	const unsigned char* vbase = (const unsigned char*)a.vc + ((size_t)kvh * a.seq_len * head_dim + d0) * EB;
	const size_t rstride = (size_t)head_dim * EB;

	const int kv_len = a.ts->kv_len, kv_pos = a.ts->kv_pos;
	const unsigned tag = fuse_tag(a.ts, f);
#ifdef CALM_TIMELINE
	unsigned long long tl[8];
	tl[0] = wall_clock64();
#endif
	// every cached row of the kv head, at once; rows past the live range are clamped into it and masked below, and so is the slot
	// this token's row is being written to (read from its granules instead)
	Raw kw[UA], vw[UA];
#pragma unroll
	for (int u = 0; u < UA; ++u) {
		const int t = min((u * NW + wave) * RPW + g, kv_len - 1);
		kw[u] = *(const Raw*)(kbase + (size_t)t * rstride);
		vw[u] = *(const Raw*)(vbase + (size_t)t * rstride);
	}

	const unsigned long long t_start = wall_clock64();
	bool bad = false;
	auto expired = [&]() {
		if (wall_clock64() - t_start > FUSE_WAIT_TICKS) {
			bad = true;
			if (lane == 0) {
				atomicOr(f.err, 1u);
			}
			return true;
		}
		return false;
	};
	const int npairs = head_dim >> 1; // <= 64
	const int pi = lane < npairs ? lane : npairs - 1;
	const unsigned long long* gq = f.gran + (size_t)h * head_dim + 2 * pi;
	const unsigned long long* gk = f.gran + (size_t)q_dim + (size_t)kvh * head_dim + 2 * pi;
	const int d = min((int)threadIdx.x, head_dim - 1); // the output dim of this thread in the last step
	unsigned long long* gv = f.gran + (size_t)q_dim + kv_dim + (size_t)kvh * head_dim + d;
	// scores are kept in base 2, scaled once through q: exp(s / sqrt(head_dim) - max) = exp2(s' - max'), s' = (q log2(e) / sqrt(head_dim)) . k
	// (src/infer.c:247-257; two operations per lane instead of a division and a multiplication per cached row)
	const float qscale = 1.44269504088896341f / sqrtf((float)head_dim);
	float* qs = sm_q + wave * (LPR * 8);

	// q: one 16-byte load per lane = the pair (2 lane, 2 lane + 1) of the head, re-read until both tags match
	u32x4 qg;
	for (;;) {
		qg = load16_agent(gq);
		if (__all(qg[1] == tag && qg[3] == tag) || expired()) {
			break;
		}
		__builtin_amdgcn_s_sleep(2);
	}
#ifdef CALM_TIMELINE
	tl[1] = wall_clock64(); // q is here (and, before it, the cached rows: the queue returns in order)
#endif
	const float qa = __uint_as_float(qg[0]) * qscale, qb = __uint_as_float(qg[2]) * qscale;
	if (lane < npairs) {
		*(float2*)(qs + 2 * lane) = make_float2(qa, qb);
	}
	__builtin_amdgcn_wave_barrier(); // same wave, LDS returns in order: no workgroup barrier
	float qv[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const float qi = qs[d0 + i];
		qv[i] = dvalid ? qi : 0.f;
	}

	// ~420 vector instructions per wave from here to the barrier (they ARE this launch's critical path once q is late: 4 waves x 16
	// tiles on one CU; profiles/r06_qkv_attn.txt): every tile's dot product first, their lane-group sums stage by stage, ONE maximum
	// for the whole wave (so that the lane groups' partial sums simply add), no masking of the value rows (a masked row's weight is 0
	// and the cache holds numbers: zero-filled at prepare_hip, finite rows since)
	float s[UA];
#pragma unroll
	for (int u = 0; u < UA; ++u) {
		s[u] = dot8_raw<KVB>(kw[u], qv);
	}
	group_sum_multi<LPR>(s);
	float m = -INFINITY;
#pragma unroll
	for (int u = 0; u < UA; ++u) {
		const int t = (u * NW + wave) * RPW + g;
		s[u] = (t < kv_len && t != kv_pos) ? s[u] : -INFINITY;
		m = fmaxf(m, s[u]);
	}
	m = wave_max_dpp(m); // wave-uniform
	float l = 0.f, o[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		o[i] = 0.f;
	}
	if (m != -INFINITY) {
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			const float p = exp2_for_asm(s[u] - m); // exp2(-inf) = 0: the masked rows
			l += p;
			axpy8_raw<KVB>(p, vw[u], o);
		}
	}
	if (lane == 0) {
		sm_m[wave] = m;
	}
	if (r == 0) {
		sm_l[wave * RPW + g] = l;
	}
	{
		float4* so = (float4*)(sm_o + (wave * RPW + g) * (LPR * 8) + r * 8); // (lanes past head_dim hold zeros)
		so[0] = make_float4(o[0], o[1], o[2], o[3]);
		so[1] = make_float4(o[4], o[5], o[6], o[7]);
	}
#ifdef CALM_TIMELINE
	asm volatile("" ::"v"(o[0]), "v"(l));
	tl[5] = wall_clock64(); // this wave's old positions are folded in
#endif
	__syncthreads();
#ifdef CALM_TIMELINE
	tl[2] = wall_clock64(); // ... every wave's
	tl[3] = tl[4] = tl[2];
#endif
	if (wave * 64 < head_dim) { // one thread per output dim from here on
		float M = sm_m[0];
#pragma unroll
		for (int w = 1; w < NW; ++w) {
			M = fmaxf(M, sm_m[w]);
		}
		float L = 0.f, O = 0.f;
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const float e = (sm_m[w] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(sm_m[w] - M); // a wave without positions
			float lw = 0.f, ow = 0.f;
#pragma unroll
			for (int gg = 0; gg < RPW; ++gg) {
				lw += sm_l[w * RPW + gg];
				ow += sm_o[(w * RPW + gg) * (LPR * 8) + d];
			}
			L = fmaf(lw, e, L);
			O = fmaf(ow, e, O);
		}
		// this token's own row: k as pairs against the q pair of the poll above (a wave-wide dot product), v one granule per thread
		// (Round 6 also tried leaving the value row to k_attn_out -- out = part without v_new + weight * v_new, the second term added
		// from the cache while that kernel stages the vector -- so that no head waits for the v rows, the row engine's last tasks:
		// k_qkv_attn - 0.1 us, k_attn_out + 0.4 us, the token slower; profiles/r06_qkv_attn.txt.)
		u32x4 kg;
		unsigned long long vg;
		for (;;) {
			vg = __hip_atomic_load(gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			kg = load16_agent(gk);
			if (__all(kg[1] == tag && kg[3] == tag && (unsigned)(vg >> 32) == tag) || expired()) {
				break;
			}
			__builtin_amdgcn_s_sleep(1);
		}
#ifdef CALM_TIMELINE
		tl[3] = wall_clock64(); // this token's k / v rows are here
#endif
		const float s_new = wave_sum(lane < npairs ? fmaf(qa, __uint_as_float(kg[0]), qb * __uint_as_float(kg[2])) : 0.f);
		const float v_new = __uint_as_float((unsigned)vg);
		const float M2 = fmaxf(M, s_new);
		const float e1 = (M == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(M - M2), pn = __builtin_amdgcn_exp2f(s_new - M2);
		float res = fmaf(pn, v_new, O * e1) / fmaf(L, e1, pn);
		if (bad) {
			res = __builtin_nanf("");
		}
		if ((int)threadIdx.x < head_dim) {
			f.out[h * head_dim + threadIdx.x] = res;
		}
#ifdef CALM_TIMELINE
		tl[4] = wall_clock64();
#endif
	}
#ifdef CALM_TIMELINE
	{
		const unsigned wv = blockIdx.x * NW + wave;
		if (lane == 0 && calm_tl_buf && wv < calm_tl_waves) {
			unsigned long long* ob = calm_tl_buf + (size_t)wv * 8;
			ob[0] = tl[0], ob[1] = tl[1], ob[2] = tl[2], ob[3] = tl[4], ob[4] = tl[3], ob[5] = tl[5];
		}
	}
#endif
}

// n_attn (= n_heads) rides in front as a preloaded scalar: a wave knows its role before it has fetched anything
template <int DB, int KVB, bool HALF, int LPR>
__global__ __launch_bounds__(WG_THREADS) void k_qkv_attn(const float* x, const float* norm_w, int dim, int q_dim, int kv_dim, int n_attn, QkvArgs a, FuseArgs f) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	if ((int)blockIdx.x < n_attn) {
		attn_fused_role<KVB, LPR>(a, f, q_dim, kv_dim, smem);
		return;
	}
	qkv_rows<DB, KVB, 4, true, HALF, false, true>(x, norm_w, dim, q_dim, kv_dim, a, &f, (int)blockIdx.x - n_attn, (int)gridDim.x - n_attn, smem);
}

// Long-context attention: one workgroup (4 waves) per (kv head, group of QH query heads, kv split).
// The K/V rows of the split are loaded ONCE and used for all QH query heads that share the kv head
// (GQA): a 4096-position context is 8 kv heads x 32 splits = 256 workgroups, each walking its 128 positions in two
// rounds of 4 K + 4 V wave-loads per wave.  Always writes split partials (o, m, l per query head) for
// k_attn_merge.  Same arithmetic as k_attn.
// TWO: the split is at most two rounds long (contexts up to 32 splits x 2 rounds: 4096 positions at head size 128) -- both rounds'
// rows are asked for at once, unconditionally, and multiplied out as they land: one exposed load latency instead of two
// (8.2 -> us at 4096 positions, profiles/r03_long_context.txt).  Otherwise rounds are loaded one ahead of the arithmetic.
constexpr int ATTN_GQA_BLOCK = 256;

template <int KVB, int LPR, int QH, bool TWO>
__global__ __launch_bounds__(ATTN_GQA_BLOCK) void k_attn_gqa(const TokState* ts, const float* qin, const void* kc, const void* vc, int head_dim, int kv_mul, int seq_len, int n_split, AttnArgs a) {
	constexpr int RPW = 64 / LPR;
	constexpr int NW = ATTN_GQA_BLOCK / 64;
	constexpr int UA = 4;
	__shared__ float sm_m[QH][NW], sm_l[QH][NW];
	__shared__ float sm_o[QH][NW][LPR * 8];

	const int lane = lane_id(), wave = wave_id();
	const int qgroups = kv_mul / QH;
	const int split = blockIdx.x % n_split;
	const int qg = (blockIdx.x / n_split) % qgroups;
	const int kvh = blockIdx.x / (n_split * qgroups);
	const int h0 = kvh * kv_mul + qg * QH; // first of this block's QH query heads
	const int r = lane % LPR, g = lane / LPR;
	const bool dvalid = r * 8 < head_dim;
	const int d0 = dvalid ? r * 8 : 0;
	const int kv_len = ts->kv_len;
	const int chunk = (kv_len + n_split - 1) / n_split;
	const int t0 = split * chunk;
	const int t1 = min(kv_len, t0 + chunk);

	constexpr int EB = KVB / 8;
	const unsigned char* kbase = (const unsigned char*)kc + ((size_t)kvh * seq_len * head_dim + d0) * EB;
	const unsigned char* vbase = (const unsigned char*)vc + ((size_t)kvh * seq_len * head_dim + d0) * EB;
	const size_t rstride = (size_t)head_dim * EB;
	constexpr int STEP = NW * RPW * UA;
	using Raw = std::conditional_t<KVB == 16, u32x4, u32x2>; // 8 cached elements
	struct Round {
		Raw k[UA], v[UA];
	};
	auto load_round = [&](Round& rd, int tb) { // clamped into the live range, masked at use
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			const int t = min(tb + u * NW * RPW + g, kv_len - 1);
			rd.k[u] = *(const Raw*)(kbase + (size_t)t * rstride);
			rd.v[u] = *(const Raw*)(vbase + (size_t)t * rstride);
		}
	};
	// the cached rows go out before anything else this wave needs (q, below, is an L2 hit and small)
	Round ra, rb;
	if constexpr (TWO) {
		load_round(ra, t0 + wave * RPW);
		load_round(rb, t0 + wave * RPW + STEP);
	}

	float qv[QH][8];
#pragma unroll
	for (int q = 0; q < QH; ++q) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			float qi = qin[(h0 + q) * head_dim + d0 + i];
			qv[q][i] = dvalid ? qi : 0.f;
		}
	}
	const float inv_sqrt_hd = 1.0f / sqrtf((float)head_dim); // one rounding away from the reference's division (src/infer.c:247)
	float m[QH], l[QH], o[QH][8];
#pragma unroll
	for (int q = 0; q < QH; ++q) {
		m[q] = -INFINITY, l[q] = 0.f;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			o[q][i] = 0.f;
		}
	}

	// one round: decode the raw rows, then per query head scores -> running (max, sum, out)
	struct Decoded {
		float kf[UA][8], vf[UA][8];
	};
	auto decode = [&](Decoded& dc, const Round& rd) {
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			if constexpr (KVB == 16) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					dc.kf[u][2 * i] = half_bits_to_float((unsigned short)(rd.k[u][i] & 0xffff));
					dc.kf[u][2 * i + 1] = half_bits_to_float((unsigned short)(rd.k[u][i] >> 16));
					dc.vf[u][2 * i] = half_bits_to_float((unsigned short)(rd.v[u][i] & 0xffff));
					dc.vf[u][2 * i + 1] = half_bits_to_float((unsigned short)(rd.v[u][i] >> 16));
				}
			} else {
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					f32x2 k0 = bf8x2_lo(rd.k[u][i]), k1 = bf8x2_hi(rd.k[u][i]);
					f32x2 v0 = bf8x2_lo(rd.v[u][i]), v1 = bf8x2_hi(rd.v[u][i]);
					dc.kf[u][4 * i] = k0[0], dc.kf[u][4 * i + 1] = k0[1], dc.kf[u][4 * i + 2] = k1[0], dc.kf[u][4 * i + 3] = k1[1];
					dc.vf[u][4 * i] = v0[0], dc.vf[u][4 * i + 1] = v0[1], dc.vf[u][4 * i + 2] = v1[0], dc.vf[u][4 * i + 3] = v1[1];
				}
			}
		}
	};
	auto accumulate = [&](const Decoded& dc, int tb) {
		bool valid[UA];
#pragma unroll
		for (int u = 0; u < UA; ++u) {
			valid[u] = tb + u * NW * RPW + g < t1;
		}
#pragma unroll
		for (int q = 0; q < QH; ++q) {
			float sc[UA];
#pragma unroll
			for (int u = 0; u < UA; ++u) {
				float d = 0.f;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					d = fmaf(qv[q][i], dc.kf[u][i], d);
				}
				d = group_sum<LPR>(d);
				sc[u] = valid[u] ? d * inv_sqrt_hd : -INFINITY;
			}
			float mn = m[q];
#pragma unroll
			for (int u = 0; u < UA; ++u) {
				mn = fmaxf(mn, sc[u]);
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
					float p = valid[u] ? __expf(sc[u] - mn) : 0.f;
					l[q] += p;
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						o[q][i] = fmaf(p, dc.vf[u][i], o[q][i]);
					}
				}
				m[q] = mn;
			}
		}
	};

	if constexpr (TWO) {
		Decoded dc;
		decode(dc, ra);
		accumulate(dc, t0 + wave * RPW);
		if (t0 + wave * RPW + STEP < t1) { // wave-uniform
			decode(dc, rb);
			accumulate(dc, t0 + wave * RPW + STEP);
		}
	} else {
		// Rounds are loaded one ahead: the raw rows of round n+1 are in flight while round n is multiplied out.  (With the
		// loads at the top of each round and one 8-wave workgroup per CU -- 160 VGPRs -- nothing hid the load latency:
		// 32.5 us per layer at a 32k context; round-ahead loads and 4-wave workgroups: 27.1, 8k: 19.7 -> 12.8.  TWO rounds ahead,
		// every load unconditional so that the waits are counted: 24.4 us at 32k against 24.3, 17.6 against 16.3 at 16k -- the
		// kernel is bound by its lane arithmetic there, not by what is in flight; not kept.)
		if (t0 + wave * RPW < t1) { // wave-uniform
			load_round(ra, t0 + wave * RPW);
		}
		for (int tb = t0 + wave * RPW; tb < t1; tb += STEP) {
			Decoded dc;
			decode(dc, ra);
			if (tb + STEP < t1) { // wave-uniform
				load_round(ra, tb + STEP);
			}
			accumulate(dc, tb);
		}
	}

	// merge lane groups, then waves, per query head
#pragma unroll
	for (int q = 0; q < QH; ++q) {
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
	// one thread per (query head, output dim) folds the NW wave partials (see k_attn)
	for (int idx = threadIdx.x; idx < QH * head_dim; idx += ATTN_GQA_BLOCK) {
		const int q = idx / head_dim, d = idx % head_dim;
		float M = sm_m[q][0];
#pragma unroll
		for (int w = 1; w < NW; ++w) {
			M = fmaxf(M, sm_m[q][w]);
		}
		float L = 0.f, O = 0.f;
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const float e = (sm_m[q][w] == -INFINITY) ? 0.f : __expf(sm_m[q][w] - M);
			L = fmaf(sm_l[q][w], e, L);
			O = fmaf(sm_o[q][w][d], e, O);
		}
		float* p = a.partial + ((size_t)(h0 + q) * n_split + split) * (head_dim + 2);
		p[d] = O;
		if (d == 0) {
			p[head_dim] = M;
			p[head_dim + 1] = L;
		}
	}
}

// ---- the same split attention on the matrix cores (head size 128), V read TRANSPOSED ------------------------------------
// k_attn_gqa is bound by its lane arithmetic from a few thousand positions on (65.6 MB of an fp8 cache at 32k in 24 us: 2.7 TB/s,
// every score a 16-lane DPP reduction).  Here a WAVE owns a tile of keys and all QH query heads of the kv head at once:
//   S^T[key][query] = K[key][:] . q[query][:]      v_mfma_f32_16x16x32_f16: A = K rows (binary16 as cached; e5m2 widened, exact),
//                                                  B = q as hi + lo binary16 (the fp32 query to 22 significant bits; columns
//                                                  beyond QH are zero), two 16-key row blocks x four k-steps of 32 dims
//   online softmax down the columns: a lane holds 8 keys of ONE query (C layout: column = lane & 15, rows 4 (lane >> 4) + e), so
//   maximum and sum are in-lane reductions plus two exchanges (lane ^ 16, lane ^ 32)
//   O^T[d][query] += V^T[d][key] . P[key][query]   A = V^T, B = P as hi + lo binary16 taken from S^T's accumulator registers as they are
// Both A operands want 8 consecutive elements of the K dimension per lane: for S that is a piece of a cached K row, for PV it is 8
// consecutive POSITIONS of one head dimension -- a piece of a row of the TRANSPOSED value cache, which this backend keeps beside
// the [position][dim] one for head size 128 ([kv head][dim][position], written by the same epilogues: k_qkv, the prompt GEMM).
// (Round 3 first transposed V through LDS -- 64 two-byte stores per lane and tile -- and ended no faster than the lane arithmetic;
// with that transposition faked the same kernel ran 8.1 instead of 10.7 us at 4096 positions, 15.1 instead of 23.7 at 32k:
// profiles/r03_long_context.txt.)  The keys of a tile are ORDERED so that the 8 scores a lane holds (C rows 4 kb + e of both row
// blocks) are 8 consecutive positions: A row n of block rb is key 8 NT (n >> 2) + 8 j + 4 rb + (n & 3) of the tile, j = the
// sub-tile.  fp16 cache: tiles of 32 keys (NT = 1); e5m2 cache: tiles of 64 keys = two sub-tiles of 32 (NT = 2), so that a lane's
// 16-byte piece of a V^T row (16 positions) serves both.  Every product is exact in fp32, accumulation is fp32; the result agrees
// with k_attn_gqa to fp32 rounding (same tolerance in the tests).  The four waves of a workgroup take different tiles of the split
// (wave-private LDS images of the RAW cache bytes, 8 KiB each for K and V^T: no barrier in the loop; a wave's LDS operations execute
// in order), rows are fetched coalesced one tile ahead (K: whole rows; V^T: 64-byte pieces of 16 rows per wave-load), and the wave
// states are folded through LDS at the end exactly as in k_attn_gqa (same partial format, same k_attn_merge).  The split length is
// rounded up to whole 64-position blocks (a V^T piece must start on a 16-byte boundary); trailing splits may be empty.
// LDS per wave: K [TILE keys][CPR 16-byte chunks], chunk index XOR a function of the key that is injective over the 16 keys one
// operand fetch touches; V^T [128 d][4 chunks], chunk index XOR ((d >> 2) & 3): conflict-free operand fetches.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void att_split2(float a, float b, unsigned& hi, unsigned& lo) { // x = hi + lo, binary16 each (|x| < 65504)
	const __half2 h = __floats2half2_rn(a, b);
	const float2 hf = __half22float2(h);
	hi = __builtin_bit_cast(unsigned, h);
	lo = __builtin_bit_cast(unsigned, __floats2half2_rn(a - hf.x, b - hf.y));
}

// QHM (4 / 8): columns of the 16-wide MFMA tile a workgroup may use; `qh` <= QHM query heads of one kv head share a workgroup (all
// kv_mul of them when kv_mul <= 8: DBRX's 6, Yi-34B's 7 -- each K / V tile is then read once per kv head; columns beyond qh are zero).
// The fp32 query is split into hi + lo binary16 AFTER a per-head power-of-two scaling that brings its largest component to 2^14
// (exact; undone, together with 1 / sqrt(head size), on the scores): any finite query fits, and small components keep their lo part.
// (Round 4 also built the merge INTO this kernel -- partial rows written through with sc1 stores, drained, an arrival counter per
// head group, the last of its n_split workgroups folding all partials read past the caches: the guide's hand-off recipe -- and
// measured it against the two launches: 14.0 against 9.1 us per layer at 4096 positions, 26.9 against 20.4 at 32k
// (profiles/r04_long_context.txt).  Write-through + drain, the counted arrival and the dependent read of 66 KB of partials are
// three memory round trips in a row on one workgroup; the merge launch and its boundary are cheaper.  Removed.)
constexpr int ATTN_MAX_SPLIT = 64;
constexpr int ATTN_VT_PSTRIDE = 128 + 4; // partial row: o[128], m, l, 2 pad -- rows stay 16-byte aligned

template <int KVB, int QHM>
__global__ __launch_bounds__(256) void k_attn_vt(const TokState* ts, const float* qin, const void* kc, const void* vt, int head_dim, int kv_mul, int seq_len, int n_split, int qh, AttnArgs a) {
	constexpr int HD = 128, NW = 4;
	constexpr int EB = KVB / 8;
	constexpr int NT = KVB == 16 ? 1 : 2, TILE = 32 * NT; // keys per tile (8 KiB of K, 8 KiB of V^T either way)
	constexpr int CPR = HD * EB / 16;                     // 16-byte chunks per K row: 16 / 8
	constexpr int RPL = 64 / CPR;                         // K rows per wave-load: 4 / 8
	constexpr int PS = ATTN_VT_PSTRIDE;
	__shared__ u32x4 kst[NW][512];
	__shared__ u32x4 vst[NW][512];
	__shared__ float sm_m[QHM][NW], sm_l[QHM][NW];
	__shared__ __attribute__((aligned(16))) float sm_o[QHM][NW][HD + 4]; // (+ 4: the heads' rows start 16 banks apart)

	const int lane = lane_id(), wave = wave_id();
	const int qgroups = kv_mul / qh;
	const int split = blockIdx.x % n_split;
	const int qg = (blockIdx.x / n_split) % qgroups;
	const int kvh = blockIdx.x / (n_split * qgroups);
	const int h0 = kvh * kv_mul + qg * qh;
	const int n = lane & 15, kb = lane >> 4; // MFMA column (query) / row (key, d) index, k-block = C row group
	const int kv_len = ts->kv_len;
	const int chunk = ((kv_len + n_split - 1) / n_split + 63) & ~63;
	const int t0 = split * chunk;
	const int t1 = min(kv_len, t0 + chunk);

	// the key's swizzle: its place (0..15) among the 16 keys of one operand fetch -- key = 8 NT a + 8 j + 4 rb + b -> 4 a + b; an e5m2
	// row is 128 bytes, so rows of even and odd b already differ in their bank phase and 2 a + (b >> 1) is enough
	auto kswz = [](int key) { return KVB == 16 ? (((key >> 3) << 2) | (key & 3)) : (((key >> 4) << 1) | ((key >> 1) & 1)); };
	const int c = lane % CPR, rr = lane / CPR; // K fetch: chunk of the row, row of the wave-load
	const int vc4 = lane & 3, vr = lane >> 2;  // V^T fetch: 16-byte piece of the 64 bytes, d row of the wave-load
	const unsigned char* kbase = (const unsigned char*)kc + (size_t)kvh * seq_len * HD * EB + c * 16;
	const unsigned char* vbase = (const unsigned char*)vt + (size_t)kvh * seq_len * HD * EB + vr * VT_BLOCK_BYTES + vc4 * 16; // (block 0 of the kv head, dim vr)
	const size_t rstride = (size_t)HD * EB;
	u32x4 kreg[8], vreg[8];
	auto fetch = [&](int tb) { // K rows tb .. tb + TILE - 1, clamped into the live range (masked at use); V^T positions tb .. tb + TILE - 1 of every d
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int row = min(tb + RPL * i + rr, kv_len - 1);
			kreg[i] = *(const u32x4*)(kbase + (size_t)row * rstride);
			vreg[i] = *(const u32x4*)(vbase + (size_t)(tb / TILE) * (HD * VT_BLOCK_BYTES) + i * (16 * VT_BLOCK_BYTES)); // a tile = one block: 8 KiB in a row
		}
	};
	auto stage = [&]() {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int key = RPL * i + rr, d = 16 * i + vr;
			kst[wave][key * CPR + (c ^ kswz(key))] = kreg[i];
			vst[wave][d * 4 + (vc4 ^ ((d >> 2) & 3))] = vreg[i];
		}
	};
	// e5m2 -> binary16: the byte becomes the upper byte (exact); two cached dwords -> four
	auto widen = [](unsigned w0, unsigned w1) -> u32x4 {
		return (u32x4){__builtin_amdgcn_perm(w0, w0, 0x050c040cu), __builtin_amdgcn_perm(w0, w0, 0x070c060cu), __builtin_amdgcn_perm(w1, w1, 0x050c040cu),
		               __builtin_amdgcn_perm(w1, w1, 0x070c060cu)};
	};
	// the 16-byte chunk `chunk` of row `row` (cpr chunks per row, chunk swizzle `swz`) of an image: 8 binary16 = one MFMA operand, or
	// 16 e5m2 = two (low / high 8 bytes, widened)
	auto chunk16 = [](const u32x4* img, int row, int cpr, int chunk, int swz) -> u32x4 { return img[row * cpr + (chunk ^ swz)]; };
	const int tb0 = t0 + wave * TILE; // this wave's tiles: tb0, tb0 + NW * TILE, ...
	if (tb0 < t1) {                   // wave-uniform
		fetch(tb0);
	}

	// the queries as B operands: column n = head h0 + n; k-step t, k-block kb, element e is head dimension 32 t + 8 kb + e for a binary16
	// cache and 64 (t >> 1) + 16 kb + 8 (t & 1) + e for an e5m2 one (a lane's 16-byte chunk of a K row then holds its operands of two
	// k-steps; the order of the summation over the head dimension is free as long as K and q agree)
	u32x4 qh_[4], ql_[4];
	float post; // what a score of this lane's query column is multiplied by: 1 / (sqrt(head size) * the query's scaling)
	{
		float4 q0[4], q1[4];
		float mx = 0.f;
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			const float* qsrc = qin + (size_t)(h0 + (n < qh ? n : 0)) * HD + (KVB == 16 ? 32 * t + 8 * kb : 64 * (t >> 1) + 16 * kb + 8 * (t & 1));
			q0[t] = *(const float4*)qsrc, q1[t] = *(const float4*)(qsrc + 4);
			mx = fmaxf(mx, fmaxf(fmaxf(fabsf(q0[t].x), fabsf(q0[t].y)), fmaxf(fabsf(q0[t].z), fabsf(q0[t].w))));
			mx = fmaxf(mx, fmaxf(fmaxf(fabsf(q1[t].x), fabsf(q1[t].y)), fmaxf(fabsf(q1[t].z), fabsf(q1[t].w))));
		}
		mx = fmaxf(mx, __shfl_xor(mx, 16)); // the head's 128 dims sit in the four lanes n, n + 16, n + 32, n + 48
		mx = fmaxf(mx, __shfl_xor(mx, 32));
		int e = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xff); // biased exponent of the largest component (NaN / inf: 255)
		e = e < 30 ? 30 : (e > 250 ? 250 : e);
		const float up = __builtin_bit_cast(float, (unsigned)(268 - e) << 23); // 2^(14 - (e - 127)): the largest component lands in [2^14, 2^15)
		post = __builtin_bit_cast(float, (unsigned)(e - 14) << 23) * (1.0f / sqrtf((float)HD));
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			unsigned hi, lo;
			att_split2(q0[t].x * up, q0[t].y * up, hi, lo), qh_[t][0] = hi, ql_[t][0] = lo;
			att_split2(q0[t].z * up, q0[t].w * up, hi, lo), qh_[t][1] = hi, ql_[t][1] = lo;
			att_split2(q1[t].x * up, q1[t].y * up, hi, lo), qh_[t][2] = hi, ql_[t][2] = lo;
			att_split2(q1[t].z * up, q1[t].w * up, hi, lo), qh_[t][3] = hi, ql_[t][3] = lo;
			if (n >= qh) {
				qh_[t] = (u32x4){0u, 0u, 0u, 0u}, ql_[t] = (u32x4){0u, 0u, 0u, 0u};
			}
		}
	}
	f32x4 o[8];
#pragma unroll
	for (int db = 0; db < 8; ++db) {
		o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
	}
	float m = -INFINITY, l = 0.f;

	for (int tb = tb0; tb < t1; tb += TILE * NW) {
		stage(); // (this tile's bytes have landed: the loads were issued a tile ago)
		if (tb + TILE * NW < t1) { // wave-uniform
			fetch(tb + TILE * NW);
		}
#pragma unroll
		for (int j = 0; j < NT; ++j) {
			if (tb + 8 * j >= t1 && NT > 1) { // wave-uniform: the sub-tile's first key lies past the split
				break;
			}
			f32x4 s[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				const int key = 8 * NT * (n >> 2) + 8 * j + 4 * rb + (n & 3); // A row n of this block
				auto kstep = [&](int t, u32x4 k16) {
					const f16x8 kop = __builtin_bit_cast(f16x8, k16);
					s[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kop, __builtin_bit_cast(f16x8, qh_[t]), s[rb], 0, 0, 0);
					s[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kop, __builtin_bit_cast(f16x8, ql_[t]), s[rb], 0, 0, 0);
				};
				if constexpr (KVB == 16) {
#pragma unroll
					for (int t = 0; t < 4; ++t) {
						kstep(t, chunk16(kst[wave], key, CPR, 4 * t + kb, kswz(key)));
					}
				} else {
#pragma unroll
					for (int u = 0; u < 2; ++u) {
						const u32x4 w = chunk16(kst[wave], key, CPR, 4 * u + kb, kswz(key));
						kstep(2 * u, widen(w[0], w[1]));
						kstep(2 * u + 1, widen(w[2], w[3]));
					}
				}
			}
			// scores of this lane's query against keys tb + 8 NT kb + 8 j + 4 rb + e   (src/infer.c:244-248)
			float mt = -INFINITY;
			const bool ragged = tb + TILE > t1; // wave-uniform: only the split's last tile has keys to mask
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const int key = tb + 8 * NT * kb + 8 * j + 4 * rb + e;
					s[rb][e] = (!ragged || key < t1) ? s[rb][e] * post : -INFINITY;
					mt = fmaxf(mt, s[rb][e]);
				}
			}
			mt = fmaxf(mt, __shfl_xor(mt, 16));
			mt = fmaxf(mt, __shfl_xor(mt, 32));
			const float mn = fmaxf(m, mt); // finite: the sub-tile's first key (kb = 0) is inside the split
			const float cs = (m == -INFINITY) ? 0.f : __expf(m - mn);
			const bool moved = __any(mn != m); // wave-uniform: some query's running maximum changed (rare once a few tiles are in)
			float ls = 0.f;
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					s[rb][e] = __expf(s[rb][e] - mn); // masked: exp(-inf) = 0
					ls += s[rb][e];
				}
			}
			ls += __shfl_xor(ls, 16);
			ls += __shfl_xor(ls, 32);
			l = l * cs + ls;
			m = mn;
			if (moved) {
#pragma unroll
				for (int db = 0; db < 8; ++db) {
					o[db] *= cs;
				}
			}
			u32x4 ph, pl; // P as the B operand: slot 4 rb + e of k-block kb = position 8 NT kb + 8 j + 4 rb + e of the tile
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				unsigned hi, lo;
				att_split2(s[rb][0], s[rb][1], hi, lo), ph[2 * rb] = hi, pl[2 * rb] = lo;
				att_split2(s[rb][2], s[rb][3], hi, lo), ph[2 * rb + 1] = hi, pl[2 * rb + 1] = lo;
			}
#pragma unroll
			for (int db = 0; db < 8; ++db) {
				const int d = 16 * db + n;
				const u32x4 vw = chunk16(vst[wave], d, 4, kb, (d >> 2) & 3); // positions 8 NT kb .. of the tile: this sub-tile's 8 are element group j
				u32x4 v16 = KVB == 16 ? vw : widen(vw[2 * (j & (NT - 1))], vw[2 * (j & (NT - 1)) + 1]);
				if (ragged) {
					// positions past the split carry P = 0, but what the cache holds there may be anything (a slot of an earlier, longer
					// sequence; 0 x inf = NaN): clear them -- only in a split's last tile
					const int live = t1 - (tb + 8 * NT * kb + 8 * j); // this lane's positions 0 .. live - 1 are inside the split
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						v16[i] = 2 * i + 1 < live ? v16[i] : (2 * i < live ? (v16[i] & 0xffffu) : 0u);
					}
				}
				const f16x8 vop = __builtin_bit_cast(f16x8, v16);
				o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vop, __builtin_bit_cast(f16x8, ph), o[db], 0, 0, 0);
				o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vop, __builtin_bit_cast(f16x8, pl), o[db], 0, 0, 0);
			}
		}
	}

	// wave states -> LDS: O^T[d = 16 db + 4 kb + e][query n]
	if (n < qh) {
		if (kb == 0) {
			sm_m[n][wave] = m;
			sm_l[n][wave] = l;
		}
#pragma unroll
		for (int db = 0; db < 8; ++db) {
			*(f32x4*)&sm_o[n][wave][16 * db + 4 * kb] = o[db];
		}
	}
	__syncthreads();
	// one thread per (query head, four output dims) folds the NW wave partials (see k_attn); a split without positions files
	// (-inf, 0, 0).  Rows of PS floats: o[HD], m, l.
	for (int idx = threadIdx.x; idx < qh * (HD / 4); idx += 256) {
		const int q = idx / (HD / 4), d = (idx % (HD / 4)) * 4;
		float M = sm_m[q][0];
#pragma unroll
		for (int w = 1; w < NW; ++w) {
			M = fmaxf(M, sm_m[q][w]);
		}
		float L = 0.f;
		f32x4 O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const float e = (sm_m[q][w] == -INFINITY) ? 0.f : __expf(sm_m[q][w] - M);
			L = fmaf(sm_l[q][w], e, L);
			O += *(const f32x4*)&sm_o[q][w][d] * e;
		}
		float* p = a.partial + ((size_t)(h0 + q) * n_split + split) * PS;
		*(f32x4*)(p + d) = O;
		if (d == 0) {
			*(f32x2*)(p + HD) = (f32x2){M, L};
		}
	}
}

// merge the kv splits of every head: grid = n_heads, one thread per output dim (block = head_dim rounded up to whole waves),
// n_split <= NS <= 64.  ONE round trip: every thread asks at once for its column of the partials (NS loads; the few indices past
// n_split re-read the last one and get weight 0), and lane s of every wave for split s's (m, l) pair -- one wave-load; the weights
// exp(m_s - M) are then one exponential per lane and reach the fold through v_readlane.  (First form: wave 0 fetched the (m, l)
// pairs, a barrier, then eight partial loads at a time: three dependent round trips, 5.1 us per launch for 130 KB --
// profiles/r03_long_context.txt.)
template <int NS>
__global__ __launch_bounds__(512) void k_attn_merge(const float* partial, float* out, int head_dim, int n_split, int stride) {
	const int h = blockIdx.x, d = threadIdx.x, lane = lane_id();
	const float* p = partial + (size_t)h * n_split * stride;
	const int dc = d < head_dim ? d : 0;
	float v[NS];
#pragma unroll
	for (int s = 0; s < NS; ++s) {
		const int sc = s < n_split ? s : n_split - 1;
		v[s] = p[sc * stride + dc];
	}
	const int sl = lane < n_split ? lane : n_split - 1;
	const float2 ml = *(const float2*)(p + sl * stride + head_dim); // (head_dim is even: 8-byte aligned)
	const float ms = lane < n_split ? ml.x : -INFINITY;
	const float M = wave_max(ms);
	const float w = ms == -INFINITY ? 0.f : __expf(ms - M); // (a split without positions, or no split at all)
	const float L = wave_sum(ml.y * w);
	float acc = 0.f;
#pragma unroll
	for (int s = 0; s < NS; ++s) {
		const float ws = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), s));
		acc = fmaf(v[s], ws, acc);
	}
	if (d < head_dim) {
		out[h * head_dim + d] = acc / L;
	}
}

// ---- attention output projection + residual:  x += wo . att      (src/infer.c:408-415) ---------
// ONE: one row per task (tiles of the same size: twice as deep) -- for row counts that leave a full grid's last round of row PAIRS
// half empty (DBRX's 6144 rows = 3072 pairs over 2048 waves; the host decides: rows_balance)
constexpr int CALM_ONE_U = 2;
// GATE (mixture-of-experts models; "forms" 8 turns it off): the router's logits for the FFN that follows are LINEAR in the residual this
// kernel completes -- logit_e = rsqrt(mean(x^2) + eps) * sum_j moegate[e][j] g[j] x[j] (src/infer.c:183-207 then :422-424) -- so the
// lane that writes x[j] also adds x[j] * (moegate[e][j] g[j]) for every expert e (lane e of the wave: one coalesced load of row j of
// the [dim][EP] fp32 table `gate_mt` that prepare_hip derives from moegate and the FFN norm weight, k_gate_prep), x[j]^2 and x[j];
// the workgroup's sums go to column blockIdx.x of `gate_part` ([EP + 2][GATE_COLS]: EP logit partials, sum of squares, sum).
// k_ffn_up folds the columns in a fixed order and knows its experts before it has seen the vector (k_ffn_up, MOE == 2): the
// routing left that kernel's critical path, where it stood between the launch and the first weight byte.
constexpr int GATE_COLS = 1024; // workgroups of k_attn_out the partial buffer has columns for
constexpr int GATE_MAX_E = 64;  // experts (one per lane)
template <int DB, int V, bool FULL, bool ONE, bool GATE, bool XREG = false>
__global__ __launch_bounds__(WG_THREADS) void k_attn_out(float* x, const float* att, const void* wo, int dim, int q_dim, const float* gate_mt, float* gate_part, int ep) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	constexpr int NR = ONE ? 1 : KShape<DB, KS_ATTN_OUT>::NR, U = ONE ? CALM_ONE_U * KShape<DB, KS_ATTN_OUT>::U : KShape<DB, KS_ATTN_OUT>::U;
	float4* xs4 = (float4*)smem;
	float* red = (float*)(xs4 + xs_slots<DB>(q_dim));
	const size_t row_bytes = (size_t)q_dim * DB / 8;
	const int lane = lane_id();
	auto rows_of = [&](int t, const unsigned char*(&rows)[NR]) {
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			rows[r] = (const unsigned char*)wo + (size_t)(t * NR + r) * row_bytes;
		}
	};
	StageRegs<V, false> sr;
	auto pre = [&]() { stage_load<WG_THREADS>(sr, att, nullptr); stage_first_barrier(); };
	auto stage = [&]() { stage_finish<DB, WG_THREADS>(sr, xs4, red, att, nullptr, q_dim, 0.f, false, nullptr); };
	float mt[NR];                       // GATE: this lane's expert's table entry of each row of the task
	float gacc = 0.f, ss = 0.f, sx = 0.f; // GATE: lane e: logit partial of expert e; every lane: sum of squares, sum
	const int el = GATE ? lane & (ep - 1) : 0;
	auto aux_of = [&](int t, float(&aux)[NR]) { // the residual values this task adds to
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			aux[r] = x[t * NR + r];
			if constexpr (GATE) {
				mt[r] = gate_mt[(size_t)(t * NR + r) * ep + el];
			}
		}
	};
	auto epi = [&](int t, float(&acc)[NR], float(&aux)[NR]) {
		if constexpr (GATE) {
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				const float xn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, aux[r] + acc[r]), RED_LANE));
				gacc = fmaf(xn, mt[r], gacc);
				ss = fmaf(xn, xn, ss);
				sx += xn;
				if (lane == RED_LANE) {
					x[t * NR + r] = xn;
				}
			}
		} else if (lane == RED_LANE) {
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				x[t * NR + r] = aux[r] + acc[r];
			}
		}
	};
	run_rows<DB, NR, U, FULL, xreg_chunks<DB, XREG>()>(dim / NR, blockIdx.x * WG_WAVES + wave_id(), gridDim.x * WG_WAVES, q_dim, xs4, att, rows_of, pre, stage, aux_of, epi);
	if constexpr (GATE) {
		// the waves' sums -> LDS -> one column of the partial buffer, waves added in index order
		constexpr int W = GATE_MAX_E + 2;
		const int wave = wave_id();
		if (lane < ep) {
			red[wave * W + lane] = gacc;
		}
		if (lane == 0) {
			red[wave * W + GATE_MAX_E] = ss;
			red[wave * W + GATE_MAX_E + 1] = sx;
		}
		__syncthreads();
		const int q = threadIdx.x;
		if (q < ep + 2) {
			const int src = q < ep ? q : GATE_MAX_E + (q - ep);
			float v = red[src];
#pragma unroll
			for (int w = 1; w < WG_WAVES; ++w) {
				v += red[w * W + src];
			}
			gate_part[(size_t)q * GATE_COLS + blockIdx.x] = v;
		}
	}
}

// gate_mt of one layer (prepare_hip): mt[j][e] = moegate[e][j] * g[j] for e < n_experts, 0 up to ep (a power of two); behind the
// table, c[e] = sum_j mt[j][e] -- what a LayerNorm's mean takes off every logit.  grid = ep + ceil(dim / 256) workgroups of 256:
// the first ep each sum one expert's column in a fixed order, the others fill the table.
template <int DB>
__global__ __launch_bounds__(256) void k_gate_prep(float* mt, const void* moegate, const float* g, int dim, int n_experts, int ep) {
	__shared__ float red[4];
	const int b = blockIdx.x;
	if (b < ep) {
		float s_ = 0.f;
		if (b < n_experts) {
			for (int j = threadIdx.x; j < dim; j += 256) {
				s_ += decode_elem<DB>(moegate, (size_t)b * dim + j) * g[j];
			}
		}
		const float t = block_sum<256>(s_, red);
		if (threadIdx.x == 0) {
			mt[(size_t)dim * ep + b] = t;
		}
		return;
	}
	const int j = (b - ep) * 256 + threadIdx.x;
	if (j < dim) {
		for (int e = 0; e < ep; ++e) {
			mt[(size_t)j * ep + e] = e < n_experts ? decode_elem<DB>(moegate, (size_t)e * dim + j) * g[j] : 0.f;
		}
	}
}

// ---- FFN up: hb = act(w1 . xn) * (w3 . xn), with optional MoE routing --------------------------

struct FfnUpArgs {
	const float* x;      // residual stream (normed here) or, for norm_par models, the saved xb
	const float* norm_w; // nullptr => x is already normalised (norm_par)
	const void *w1, *w3, *moegate;
	float* he;      // (n_active, hidden)
	float* moe_w;   // (n_active) routing weights, written by block 0
	int* moe_e;     // (n_active) routed expert ids
	int dim, hidden, n_experts, n_active;
	float eps;
	int ln, gelu;
	int cut;             // tasks of the first-dispatched half of the grid (task_range); 0: tasks dealt round robin over all workgroups
	const float* gate_c; // MOE == 2: c[e] behind the layer's gate_mt table (k_gate_prep), what a LayerNorm's mean takes off logit e
};

// The router (src/infer.c:277-305) by one wave, one expert per lane (n_experts <= 64): the experts in the order (larger logit
// first, equal logits: lower index first), the first n_active of them taken, then the softmax over the winners only, their
// exponentials summed in rank order; rank k's expert and weight end up in sel_e[k] / sel_w[k] (and, if out_w, there).
// A lane's rank is the number of experts ahead of it, counted against every other lane's logit through v_readlane (a scalar
// broadcast: no LDS crossbar round trip -- the first form ran n_active rounds of a wave-wide arg-max over ds_bpermute shuffles,
// ~50 dependent LDS round trips for Mixtral's 2 of 8, ~90 for DBRX's 4 of 16: 2-5 us at the head of every k_ffn_up launch).
// Every wave of a workgroup may call it with the same logits (identical stores to sel_e / sel_w; a wave then reads what it wrote).
__device__ __forceinline__ void moe_route(float logit, int n_experts, int n_active, int* sel_e, float* sel_w, float* out_w, int* out_e) {
	const int lane = lane_id();
	const bool valid = lane < n_experts;
	logit = valid ? logit : 0.f;
	int rank = 0;
	for (int j = 0; j < n_experts; ++j) {
		const float lj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, logit), j));
		rank += (lj > logit || (lj == logit && j < lane)) ? 1 : 0;
	}
	float top = 0.f, ex = 0.f;
	int pick = 0;
	for (int k = 0; k < n_active; ++k) {
		const unsigned long long mask = __ballot(valid && rank == k);
		const int idx = mask ? (int)__builtin_ctzll(mask) : min(k, n_experts - 1); // (no lane of that rank: logits that do not order -- NaN)
		const float lk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, logit), idx));
		top = k == 0 ? lk : top; // the largest logit overall: the softmax's reference point
		if (lane == k) {
			pick = idx;
			ex = expf(lk - top);
		}
	}
	float denom = 0.f;
	for (int k = 0; k < n_active; ++k) {
		denom += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), k));
	}
	if (lane < n_active) {
		sel_e[lane] = pick;
		sel_w[lane] = ex / denom;
		if (out_w) {
			out_w[lane] = ex / denom;
			out_e[lane] = pick;
		}
	}
}

// (Round 4: v_exp_f32 + v_rcp_f32 in place of libm's expf and the IEEE division -- ~45 of the ~150 instructions of a k_ffn_up task's
// epilogue -- measured the same to the tenth of a microsecond: profiles/r04_startup.txt.  The exact form stays.)
__device__ __forceinline__ float act_silu(float x) {
	return x / (1.0f + expf(-x)); // src/infer.c:273-275
}
__device__ __forceinline__ float act_gelu(float x) {
	return 0.5f * x * (1.0f + tanhf(0.797885f * (x + 0.044715f * x * x * x))); // src/infer.c:269-271
}

// task = one hidden unit j of one active expert slot k: rows (w1[e_k][j], w3[e_k][j]) [x2 for gf4]
// MOE: 0 dense; 1 the gate computed here, by every workgroup, from the vector (below); 2 the gate folded from the partials
// k_attn_out's epilogue left (`moegate` then points at gate_part, `n_experts` carries n_experts | k_attn_out's grid << 8 | its expert rows ep << 24).
template <int DB, int V, bool FULL, int MOE, bool XREG = false>
__global__ __launch_bounds__(WG_THREADS) void k_ffn_up(const float* x, const float* norm_w, const void* w1, const void* w3, const void* moegate, int dim, int hidden, int n_experts, int n_active,
                                                 FfnUpArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	constexpr int NR = KShape<DB, KS_FFN_UP>::NR, U = KShape<DB, KS_FFN_UP>::U;
	constexpr int JP = NR / 2; // hidden units per task
	float4* xs4 = (float4*)smem;
	float* red = (float*)(xs4 + xs_slots<DB>(dim));
	float* gate = red + 16;                   // n_experts logits (MOE == 2: + sum of squares, sum)
	float* sel_w = gate + GATE_MAX_E + 2;     // n_active
	int* sel_e = (int*)(sel_w + GATE_MAX_E);  // n_active
	const size_t row_bytes = (size_t)dim * DB / 8;
	const int lane = lane_id(), wave = wave_id();
	const int nact = n_active > 0 ? n_active : 1;
	const int per_expert = hidden / JP;
	const int ntasks = nact * per_expert;
	constexpr bool moe = MOE != 0;

	// task t -> expert slot k = t / per_expert and hidden unit j: dense models have one slot; otherwise a float reciprocal and one
	// correction step (an integer division is ~25 scalar instructions per task, twice)
	const float inv_pe = 1.0f / (float)per_expert;
	auto split = [&](int t, int& k, int& j) {
		if constexpr (!moe) {
			k = 0, j = t * JP;
		} else {
			k = (int)(((float)t + 0.5f) * inv_pe);
			int r = t - k * per_expert;
			if (r < 0) {
				--k, r += per_expert;
			} else if (r >= per_expert) {
				++k, r -= per_expert;
			}
			j = r * JP;
		}
	};
	auto rows_of = [&](int t, const unsigned char*(&rows)[NR]) {
		int k, j;
		split(__builtin_amdgcn_readfirstlane(t), k, j);
		int e = moe ? sel_e[k] : 0;
		size_t base = ((size_t)e * hidden + j) * row_bytes;
#pragma unroll
		for (int p = 0; p < JP; ++p) {
			rows[2 * p] = (const unsigned char*)w1 + base + p * row_bytes;
			rows[2 * p + 1] = (const unsigned char*)w3 + base + p * row_bytes;
		}
	};
	auto no_aux = [&](int, float(&)[NR]) {};
	float nscale = 1.f; // what the norm leaves to the epilogue (stage_finish)
	auto epi = [&](int t, float(&acc)[NR], float(&)[NR]) {
		if (lane == RED_LANE) {
			int k, j;
			split(t, k, j);
#pragma unroll
			for (int p = 0; p < JP; ++p) {
				float u = acc[2 * p] * nscale, g = acc[2 * p + 1] * nscale;
				a.he[(size_t)k * hidden + j + p] = (a.gelu ? act_gelu(u) : act_silu(u)) * g; // src/infer.c:440-450
			}
		}
	};

	StageRegs<V, true> sr;
	if constexpr (MOE == 0) {
		auto pre = [&]() { stage_load<WG_THREADS>(sr, x, norm_w); stage_first_barrier(); };
		auto stage = [&]() { nscale = stage_finish<DB, WG_THREADS>(sr, xs4, red, x, norm_w, dim, a.eps, a.ln != 0, nullptr); };
		const TaskRange tr = task_range(ntasks, WG_WAVES, wave, a.cut);
		run_rows<DB, NR, U, FULL, xreg_chunks<DB, XREG>()>(tr.ntasks, tr.first, tr.stride, dim, xs4, x, rows_of, pre, stage, no_aux, epi);
		if (blockIdx.x == 0 && threadIdx.x == 0) {
			a.moe_w[0] = 1.0f; // src/infer.c:430-432
			a.moe_e[0] = 0;
		}
		return;
	}

	if constexpr (MOE == 2) {
		// The routing from k_attn_out's partial sums: [ne + 2][GATE_COLS] floats, `cols` live columns (one per workgroup of that
		// launch).  Quantity q (expert logits, then the sum of squares and the sum) is folded by wave q % WG_WAVES: every lane
		// adds its float4s' live columns in index order, then the wave's DPP tree -- the same order in every workgroup, so all of
		// them pick the same experts -- and every load of a wave is in flight at once (one round trip).  The vector's own
		// loads go out first (the image needs them right after); nothing here waits for the image.
		// (k_attn_out<GATE> files the two statistics behind ITS expert rows, whose count ep is n_experts rounded up to a power of two:
		// rows ep and ep + 1; quantity q of the fold lives in row q < ne ? q : ep + (q - ne))
		const int ne = n_experts & 0xff, cols = (n_experts >> 8) & 0xffff, ep = (unsigned)n_experts >> 24, nq = ne + 2;
		const float* part = (const float*)moegate;
		stage_load<WG_THREADS>(sr, x, norm_w);
		// quantities per wave and batch x float4s per lane and quantity: up to 20 loads in flight per lane whichever way -- few live
		// columns (a small model's k_attn_out grid) leave room for more quantities per round trip (64 experts: 17 per wave)
		auto fold = [&](auto QBc, auto C4c) {
			constexpr int QB = decltype(QBc)::value, C4 = decltype(C4c)::value;
			for (int qb = 0; qb < nq; qb += WG_WAVES * QB) {
				float4 v[QB][C4];
#pragma unroll
				for (int j = 0; j < QB; ++j) {
					const int q = qb + wave + WG_WAVES * j;
#pragma unroll
					for (int i = 0; i < C4; ++i) {
						v[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
						if (q < nq && i * 256 < cols) { // wave-uniform
							v[j][i] = *((const float4*)(part + (size_t)(q < ne ? q : q - ne + ep) * GATE_COLS) + i * 64 + lane);
						}
					}
				}
#pragma unroll
				for (int j = 0; j < QB; ++j) {
					const int q = qb + wave + WG_WAVES * j;
					if (q < nq) { // wave-uniform
						float s_ = 0.f;
#pragma unroll
						for (int i = 0; i < C4; ++i) {
							const int c0 = i * 256 + lane * 4;
							s_ += (c0 < cols ? v[j][i].x : 0.f) + (c0 + 1 < cols ? v[j][i].y : 0.f) + (c0 + 2 < cols ? v[j][i].z : 0.f) + (c0 + 3 < cols ? v[j][i].w : 0.f);
						}
						s_ = wave_sum63(s_);
						if (lane == RED_LANE) {
							gate[q] = s_;
						}
					}
				}
			}
		};
		if (cols <= 256) {
			fold(std::integral_constant<int, 17>(), std::integral_constant<int, 1>());
		} else if (cols <= 512) {
			fold(std::integral_constant<int, 9>(), std::integral_constant<int, 2>());
		} else {
			fold(std::integral_constant<int, 5>(), std::integral_constant<int, GATE_COLS / 256>());
		}
		__syncthreads();
		{
			// every wave routes for itself (the same logits, the same picks, identical stores): no second barrier before the first tile
			const float n = (float)dim;
			const float mean = a.ln ? gate[ne + 1] / n : 0.f;                       // src/infer.c:183-207
			// (one-pass variance from the partial sums; the reference's two-pass form cannot go negative, this one can by rounding)
			const float scale = 1.0f / sqrtf(fmaxf(gate[ne] / n - mean * mean, 0.f) + a.eps);
			const int le = lane < ne ? lane : 0;
			const float c = a.ln ? a.gate_c[le] : 0.f;
			moe_route((gate[le] - mean * c) * scale, ne, n_active, sel_e, sel_w, (blockIdx.x == 0 && wave == 0) ? a.moe_w : nullptr, a.moe_e);
		}
		auto nothing = [&]() {};
		auto stage = [&]() { nscale = stage_finish<DB, WG_THREADS>(sr, xs4, red, x, norm_w, dim, a.eps, a.ln != 0, nullptr); };
		const TaskRange tr = task_range(ntasks, WG_WAVES, wave, a.cut);
		run_rows<DB, NR, U, FULL, xreg_chunks<DB, XREG>()>(tr.ntasks, tr.first, tr.stride, dim, xs4, x, rows_of, nothing, stage, no_aux, epi);
		return;
	}

	// MoE: the routing decides which rows to stream, so it has to come first.  Every workgroup recomputes the gate
	// (n_experts short rows, L2-resident after the first workgroup) -- no cross-workgroup hand-off.  What does NOT depend
	// on the routing is asked for up front: the vector and its norm weight, then the gate rows themselves (they are weights),
	// so that the norm prologue runs while they fly.  Wave w owns experts w, w + WG_WAVES, ...; its loads are numbered
	// j = (expert slot i) * chunks + (1-KiB chunk k of the row); the first GP of them are prefetched into registers.
	constexpr int GP = 8; // (16 measured no better on the 8-expert shape: eight surplus loads per wave ahead of the weight stream)
	const int nl = dim / Fmt<DB>::G;          // 16-byte lane-loads per row
	const int chunks = (nl + 63) >> 6;          // wave-loads per row
	const int per_wave = (n_experts + WG_WAVES - 1) / WG_WAVES; // expert slots of a wave
	const int total = per_wave * chunks;
	auto gate_src = [&](int j, int& e, int& k) -> gptr16 {
		const int i = j / chunks;
		k = j - i * chunks;
		e = wave + WG_WAVES * i;
		const int ec = min(e, n_experts - 1), li = min(k * 64 + lane, nl - 1); // always in bounds; masked at use
		return (gptr16)((const unsigned char*)moegate + (size_t)ec * row_bytes) + li;
	};
	stage_load<WG_THREADS>(sr, x, norm_w);
	u32x4 gw[GP];
#pragma unroll
	for (int j = 0; j < GP; ++j) {
		int e, k;
		gw[j] = *gate_src(min(j, total - 1), e, k);
	}
	nscale = stage_finish<DB, WG_THREADS>(sr, xs4, red, x, norm_w, dim, a.eps, a.ln != 0, nullptr);
	{
		f32x2 acc2 = {0.f, 0.f};
		auto gate_step = [&](u32x4 w, int j) { // multiply-add load j; a row's last chunk reduces and files the logit
			int e, k;
			(void)gate_src(j, e, k);
			const bool live = k * 64 + lane < nl;
			w = live ? w : (u32x4){0u, 0u, 0u, 0u};
			if (k == 0) {
				acc2 = (f32x2){0.f, 0.f};
			}
			acc2 = dot16<DB>(w, (const f32x4*)xs4 + k * Fmt<DB>::CS + lane, acc2);
			if (k == chunks - 1) {
				const float logit = wave_sum63(acc2[0] + acc2[1]) * nscale;
				if (lane == RED_LANE && e < n_experts) {
					gate[e] = logit;
				}
			}
		};
#pragma unroll
		for (int j = 0; j < GP; ++j) {
			if (j < total) { // wave-uniform
				gate_step(gw[j], j);
			}
		}
		for (int j = GP; j < total; ++j) { // more gate rows than the prefetch holds (16 experts x 6 KiB rows: 24 loads per wave)
			int e, k;
			const u32x4 w = *gate_src(j, e, k);
			gate_step(w, j);
		}
		__syncthreads();
		if (wave == 0) {
			moe_route(gate[lane < n_experts ? lane : 0], n_experts, n_active, sel_e, sel_w, blockIdx.x == 0 ? a.moe_w : nullptr, a.moe_e);
		}
		__syncthreads();
	}
	auto nothing = [&]() {};
	run_rows<DB, NR, U, FULL>(ntasks, blockIdx.x * WG_WAVES + wave, gridDim.x * WG_WAVES, dim, xs4, x, rows_of, nothing, nothing, no_aux, epi);
}

// ---- FFN down + weighted residual:  x += sum_k moe_w[k] * (w2[e_k] . he[k])  (src/infer.c:452-456)
// Experts are added in rank order (k = 0, 1, ...) by the same lane, so the sum order is the
// reference's and is deterministic (the CUDA path's atomicAdd, src/infer.cu:618, is not).
// UO (tile depth override): rows whose chunk count the format's tile depth does not divide waste the surplus loads of their last
// step (a clamped re-read: no HBM bytes, but a slot of the CU's memory queue each).  UO = 2: 2 rows x 2 chunks for fp8 / fp16 rows
// of 4 n + 2 chunks (hidden 14336 at fp8 = 14: 7 exact steps instead of 4 + 4 + 4 + 2 of 4; Mistral-7B 12.9 -> 12.0 us per launch).
// UO = 7: rows of 7k KiB (hidden 14336 at fp8 = 14 chunks, fp16 = 28, gf4 = 7): tiles of 2 rows x 7 chunks, so a
// wave's first two steps -- issued before the prologue -- already cover 28 KiB, the whole task at fp8;
// the long prologue of this kernel (staging the hidden-sized vector) then hides behind the full stream.
// k0 / kn: the columns [k0, k0 + kn) of w2 this launch covers.  Normally all of them; a hidden_dim whose fp32 image does not fit
// the CU's LDS is covered by several launches over whole-KiB column ranges, each adding its partial products onto x.
// (Tried for mixture-of-experts models: asking for expert k + 1's hidden vector while expert k's rows stream, so that a later expert
// starts with a barrier and LDS stores only -- Mixtral-8x7B 25.0 us per launch against 23.1 without: the eight extra loads per
// wave sit in the queue ahead of the next tiles.  Not kept.  Round 4, two more forms of "never drain between experts", both
// bit-identical to this one and both slower (profiles/r04_moe.txt): all active experts' hidden vectors side by side in LDS with a task's
// row the concatenation of the experts' rows (Mixtral 22.4 against 21.5 us, DBRX 51 against 49: the longer prologue costs more than
// the restart it saves), and one tile stream across the expert boundary with the image swapped in-stream behind a barrier (22.2 / 58:
// the swap's loads go out only after the slowest wave has arrived, where a restart's go out as each wave finishes).)
// SEG (mixtures of many small experts; the launcher decides, ffn_down_segs): ALL active experts' hidden vectors side by side in LDS
// and a task = the same output rows of every active expert, one tile stream (run_rows_impl SEGS).  The epilogue adds expert after
// expert in rank order through one lane: bit-identical to the one-pass-per-expert form.  Lane s of every wave holds rank s's
// expert id and weight (n_active <= 64); a segment's are broadcast from there (v_readlane), no memory access on the way.
template <int DB, int BLOCK, int V, int UO, bool FULL, bool SEG>
__global__ __launch_bounds__(BLOCK) void k_ffn_down(float* x, const float* he, const void* w2, const float* moe_w, const int* moe_e, int dim, int hidden,
                                                    int n_active, int k0, int kn) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	// UO: tiles of 2 rows x UO chunks instead of the format's shape; UO = 1: ONE row x 4 chunks, for matrices of fewer row pairs than
	// the chip has waves (TinyLlama's 2048 rows: 7.9 -> 6.6 us)
	// (Round 4: every task split along K between an older and a younger wave of the workgroup -- 8 + 6 of Mistral-7B's 14 chunks, the partial sums
	// handed over through LDS: the older four waves of a workgroup leave 0.8-1.1 us before the younger four -- 12.18 -> 12.0 us, gf4 unchanged:
	// profiles/r04_startup.txt; not kept.  Round 6: s_setprio 1 / 3 for the younger four, and for k_ffn_up's later-dispatched workgroups:
	// 11.8 -> 11.8-12.0 us, k_ffn_up 19.3 -> 20.1, the token 0.9-1.5 % slower -- profiles/r06_gf4.txt.)
	// (Round 4: ONE row x 11 chunks for DBRX's ragged 10.5-KiB rows -- one exact step instead of 4 + 4 + 3 and a clamped surplus load --
	// measured 51 us against 49: profiles/r04_moe.txt.)
	// SEG: UO = chunks per step (1 / 2 / 4: the largest that divides a segment's chunk count -- a segment is walked in whole steps)
	// + 8 for ONE row per task
	constexpr int NR = SEG ? ((UO & 8) ? 1 : 2) : (UO == 1 ? 1 : (UO ? 2 : KShape<DB, KS_FFN_DOWN>::NR));
	constexpr int U = SEG ? (UO & 7) : (UO == 1 ? 4 : (UO ? UO : KShape<DB, KS_FFN_DOWN>::U));
	constexpr int NW = BLOCK / 64;
	float4* xs4 = (float4*)smem;
	const size_t row_bytes = (size_t)hidden * DB / 8;
	const int lane = lane_id();
	const int nact = n_active > 0 ? n_active : 1;
	if constexpr (SEG) {
		const int seg_slots = xs_slots<DB>(kn); // float4 slots of one expert's image
		float* red = (float*)(xs4 + nact * seg_slots);
		const int my_e = moe_e[min(lane, nact - 1)];
		const float my_w = moe_w[min(lane, nact - 1)];
		const unsigned char* const wcol = (const unsigned char*)w2 + (size_t)k0 * DB / 8;
		const size_t expert_bytes = (size_t)dim * row_bytes;
		auto rows_of = [&](int t, int seg, const unsigned char*(&rows)[NR]) {
			const unsigned char* base = wcol + (size_t)__builtin_amdgcn_readlane(my_e, seg) * expert_bytes;
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				rows[r] = base + (size_t)(t * NR + r) * row_bytes;
			}
		};
		// FULL (whole-chunk vectors): the experts' images side by side ARE the image of the concatenated vector he[0 .. nact) -- one
		// staging pass for all of them, its loads ahead of the first tiles like any dense kernel's.  Ragged vectors: the first
		// expert's goes out ahead of the tiles, the others are staged one after the other behind them.
		StageRegs<V, false> sr;
		auto pre = [&]() { stage_load<BLOCK>(sr, he + k0, nullptr); stage_first_barrier(); };
		auto stage = [&]() {
			if constexpr (FULL) {
				stage_finish<DB, BLOCK>(sr, xs4, red, he + k0, nullptr, nact * kn, 0.f, false, nullptr); // (launcher: k0 == 0, kn == hidden)
			} else {
				stage_finish<DB, BLOCK>(sr, xs4, red, he + k0, nullptr, kn, 0.f, false, nullptr);
				for (int s_ = 1; s_ < nact; ++s_) {
					const float* hk = he + (size_t)s_ * hidden + k0;
					StageRegs<V, false> s2;
					stage_load<BLOCK>(s2, hk, nullptr);
					stage_finish<DB, BLOCK>(s2, xs4 + s_ * seg_slots, red, hk, nullptr, kn, 0.f, false, nullptr);
				}
			}
		};
		auto aux_of = [&](int t, float(&aux)[NR]) { // residual so far
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				aux[r] = x[t * NR + r];
			}
		};
		float carry[NR]; // the task's running value between its segments (meaningful in lane RED_LANE)
		auto epi = [&](int t, int seg, float(&acc)[NR], float(&aux)[NR]) {
			const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), seg));
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				carry[r] = (seg == 0 ? aux[r] : carry[r]) + acc[r] * w;
			}
			if (seg == nact - 1 && lane == RED_LANE) {
#pragma unroll
				for (int r = 0; r < NR; ++r) {
					x[t * NR + r] = carry[r];
				}
			}
		};
		run_rows_segs<DB, NR, U, FULL>(dim / NR, blockIdx.x * NW + wave_id(), gridDim.x * NW, kn, nact, xs4, he, rows_of, pre, stage, aux_of, epi);
		return;
	}
	float* red = (float*)(xs4 + xs_slots<DB>(kn));
	for (int k = 0; k < nact; ++k) {
		const float wk = moe_w[k];
		const unsigned char* wbase = (const unsigned char*)w2 + (size_t)moe_e[k] * dim * row_bytes + (size_t)k0 * DB / 8;
		const float* hk = he + (size_t)k * hidden + k0;
		auto rows_of = [&](int t, const unsigned char*(&rows)[NR]) {
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				rows[r] = wbase + (size_t)(t * NR + r) * row_bytes;
			}
		};
		StageRegs<V, false> sr;
		auto pre = [&]() { stage_load<BLOCK>(sr, hk, nullptr); stage_first_barrier(); };
		auto stage = [&]() {
			if (k > 0) {
				__syncthreads(); // everyone is done reading the previous expert's image
			}
			stage_finish<DB, BLOCK>(sr, xs4, red, hk, nullptr, kn, 0.f, false, nullptr);
		};
		auto aux_of = [&](int t, float(&aux)[NR]) { // residual so far (same lane wrote it for k > 0)
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				aux[r] = x[t * NR + r];
			}
		};
		auto epi = [&](int t, float(&acc)[NR], float(&aux)[NR]) {
			if (lane == RED_LANE) {
#pragma unroll
				for (int r = 0; r < NR; ++r) {
					x[t * NR + r] = aux[r] + acc[r] * wk;
				}
			}
		};
		run_rows<DB, NR, U, FULL>(dim / NR, blockIdx.x * NW + wave_id(), gridDim.x * NW, kn, xs4, he, rows_of, pre, stage, aux_of, epi);
	}
}

// ---- final norm + classifier   (src/infer.c:465-469) -----------------------------------------
template <int DB, int V, bool FULL, bool XREG = false>
__global__ __launch_bounds__(WG_THREADS) void k_output(float* logits, const float* x, const float* norm_w, const void* wcls, int dim, int vocab, float eps, int ln, int cut) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	constexpr int NR = KShape<DB, KS_OUTPUT>::NR, U = KShape<DB, KS_OUTPUT>::U;
	float4* xs4 = (float4*)smem;
	float* red = (float*)(xs4 + xs_slots<DB>(dim));
	const size_t row_bytes = (size_t)dim * DB / 8;
	const int lane = lane_id();
	const int ntasks = (vocab + NR - 1) / NR;
	auto rows_of = [&](int t, const unsigned char*(&rows)[NR]) {
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			int j = min(t * NR + r, vocab - 1); // ragged last task: re-read the last row, drop the result
			rows[r] = (const unsigned char*)wcls + (size_t)j * row_bytes;
		}
	};
	StageRegs<V, true> sr;
	auto pre = [&]() { stage_load<WG_THREADS>(sr, x, norm_w); stage_first_barrier(); };
	float nscale = 1.f; // what the norm leaves to the epilogue (stage_finish)
	auto stage = [&]() { nscale = stage_finish<DB, WG_THREADS>(sr, xs4, red, x, norm_w, dim, eps, ln != 0, nullptr); };
	auto no_aux = [&](int, float(&)[NR]) {};
	auto epi = [&](int t, float(&acc)[NR], float(&)[NR]) {
		if (lane == RED_LANE) {
#pragma unroll
			for (int r = 0; r < NR; ++r) {
				if (t * NR + r < vocab) {
					logits[t * NR + r] = acc[r] * nscale;
				}
			}
		}
	};
	const TaskRange tr = task_range(ntasks, WG_WAVES, wave_id(), cut);
	run_rows<DB, NR, U, FULL, xreg_chunks<DB, XREG>()>(tr.ntasks, tr.first, tr.stride, dim, xs4, x, rows_of, pre, stage, no_aux, epi);
}

// ---- greedy sampler on the device: first index of the strict maximum (src/sampler.c:34-42) ----
// single workgroup of 1024 threads; writes *next and, if trace, appends to trace[(*trace_count)++]
__global__ __launch_bounds__(1024) void k_argmax(const float* logits, int n, int* next, int* trace, int* trace_count) {
	// The reference scans from (max = -FLT_MAX, index = -1) taking every strictly greater value (src/sampler.c:34-42): NaNs and
	// values <= -FLT_MAX never win.  With nothing to pick it returns -1 and its caller indexes the embedding table with that;
	// here the degenerate case yields token 0 instead (the chained decode gathers embedding row `*next` on the device).
	__shared__ float sv[16];
	__shared__ int si[16];
	float best = -3.402823466e+38f; // -FLT_MAX: values must be strictly greater to be picked
	int bi = -1;
	for (int i = threadIdx.x; i < n; i += 1024) { // ascending per thread: strict > keeps the first
		float v = logits[i];
		if (v > best) {
			best = v;
			bi = i;
		}
	}
	auto better = [](float v2, int i2, float v, int i) { return i2 >= 0 && (i < 0 || v2 > v || (v2 == v && i2 < i)); };
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		float v2 = __shfl_xor(best, o);
		int i2 = __shfl_xor(bi, o);
		if (better(v2, i2, best, bi)) {
			best = v2;
			bi = i2;
		}
	}
	if (lane_id() == 0) {
		sv[threadIdx.x >> 6] = best;
		si[threadIdx.x >> 6] = bi;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 16; ++w) {
			if (better(sv[w], si[w], best, bi)) {
				best = sv[w];
				bi = si[w];
			}
		}
		if (bi < 0) {
			bi = 0;
		}
		*next = bi;
		if (trace) {
			int slot = (*trace_count)++;
			trace[slot] = bi;
		}
	}
}

// ---- min-p sampler on the device (src/sampler.c:44-78, driven as src/sampler.c:80-90) ---------------------------------
// The host sampler keeps every token whose logit is within log(minp) * temperature of the maximum, weighs the survivors by
// exp((logit - max) / temperature), and walks their running sum in index order up to coin * total.  Here: a workgroup of 1024
// threads, thread t owning the contiguous slice of indices [t * per, (t + 1) * per) -- so that "index order" survives: partial
// sums per slice (each in index order), thread 0 walks the 1024 partials in order, then the slice that contains the draw.
// The coin comes from the sampler's xorshift* state kept on the device (src/sampler.c:7-18), one draw per sampled token.
// What differs from the host: expf is the device's (<= 1 ulp from libm's) and the running sum is bracketed per slice, so a draw
// that lands within a few ulps of a boundary between two survivors may pick the neighbour.  (The reference is compiled with
// -ffast-math: its own summation order is the compiler's choice.)
struct SampleState {
	unsigned long long rng; // src/sampler.h:5
	float temperature;
	float cutoff_offset; // logf(minp) * temperature, evaluated on the host (src/sampler.c:52)
};

__global__ __launch_bounds__(1024) void k_sample_minp(const float* logits, int n, int* next, int* trace, int* trace_count, SampleState* st) {
	__shared__ float red[16];
	__shared__ float part[1024];
	__shared__ int last[1024];
	const int t = threadIdx.x;
	const int per = (n + 1023) / 1024;
	const int i0 = min(t * per, n), i1 = min(i0 + per, n);

	float mx = -3.402823466e+38f; // src/sampler.c:46-49
	for (int i = t; i < n; i += 1024) {
		const float v = logits[i];
		mx = v > mx ? v : mx;
	}
	mx = wave_max(mx);
	if (lane_id() == 0) {
		red[t >> 6] = mx;
	}
	__syncthreads();
	mx = red[0];
#pragma unroll
	for (int w = 1; w < 16; ++w) {
		mx = fmaxf(mx, red[w]);
	}
	const float temperature = st->temperature;
	const float cutoff = mx + st->cutoff_offset; // src/sampler.c:52

	float sum = 0.f;
	int lst = -1;
	for (int i = i0; i < i1; ++i) { // src/sampler.c:58-66 over this slice
		const float v = logits[i];
		if (v >= cutoff) {
			sum += expf((v - mx) / temperature);
			lst = i;
		}
	}
	part[t] = sum;
	last[t] = lst;
	__syncthreads();
	if (t != 0) {
		return;
	}
	// the coin: xorshift* (src/sampler.c:7-18)
	unsigned long long s = st->rng;
	s ^= s >> 12;
	s ^= s << 25;
	s ^= s >> 27;
	st->rng = s;
	const unsigned u = (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 32);
	const float coin = (float)(u >> 8) / 16777216.0f;

	float total = 0.f;
	int fallback = 0;
	for (int k = 0; k < 1024; ++k) {
		total += part[k];
		fallback = last[k] >= 0 ? last[k] : fallback; // the last survivor (src/sampler.c:62)
	}
	const float r = coin * total; // src/sampler.c:69
	float cdf = 0.f;
	int pick = -1;
	for (int k = 0; k < 1024 && pick < 0; ++k) {
		if (last[k] < 0) {
			continue;
		}
		if (r < cdf + part[k]) { // the draw falls into slice k: walk it (src/sampler.c:71-76)
			const int a = min(k * per, n), b = min(a + per, n);
			for (int i = a; i < b; ++i) {
				const float v = logits[i];
				if (v >= cutoff) {
					cdf += expf((v - mx) / temperature);
					if (r < cdf) {
						pick = i;
						break;
					}
				}
			}
			if (pick < 0) {
				pick = last[k]; // rounding between the bracketed and the running sum
			}
		} else {
			cdf += part[k];
		}
	}
	if (pick < 0) {
		pick = fallback;
	}
	*next = pick;
	if (trace) {
		const int slot = (*trace_count)++;
		trace[slot] = pick;
	}
}

} // namespace calm
