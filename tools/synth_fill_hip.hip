// synth_fill_hip.hip -- FIXTURE TOOLING on the device (not the decode path): what tools/synth_fill.c and calmfile.quantize_gf4 do
// on the host, done where the weights are going to live, so that the 46.7 GB (Mixtral-8x7B fp8) and 131.6 GB (DBRX-132B fp8)
// BASELINE shapes load in seconds instead of minutes of host OpenMP + PCIe.
//
//   synth_fill_hip    : the SAME bytes as synth_fill() for the same (n, kind, lut, seed): splitmix64 is a counter hash -- draw j of
//                       chunk c is mix(seed' + c * G + (j + 1) * G) -- so every draw is independent and a thread can own it
//   quantize_gf4_hip  : the reference converter's gf4() (tools/convert.py:247-268) per group of 8 fp32 values -> one 32-bit word,
//                       bit for bit what calmfile.quantize_gf4 (and with it the reference's torch code) produces
//
// The buffers come from the product library's alloc_hip (they carry the slack its kernels' unclamped loads need); this file
// only fills them.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/libsynth_fill_hip.so tools/synth_fill_hip.hip
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                                            \
	do {                                                                                                 \
		hipError_t e_ = (x);                                                                             \
		if (e_ != hipSuccess) {                                                                          \
			fprintf(stderr, "synth_fill_hip: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			abort();                                                                                     \
		}                                                                                                \
	} while (0)

namespace {

constexpr uint64_t GOLDEN = 0x9e3779b97f4a7c15ull;
constexpr size_t CHUNK = 1 << 20; // elements per independent stream (tools/synth_fill.c)

__device__ __forceinline__ uint64_t mix64(uint64_t z) { // splitmix64's output function
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}
// draw number j (0-based) of chunk c
__device__ __forceinline__ uint64_t draw(uint64_t seed, size_t c, size_t j) {
	const uint64_t s0 = seed * 0x2545f4914f6cdd1dull + (uint64_t)c * GOLDEN + 1;
	return mix64(s0 + (uint64_t)(j + 1) * GOLDEN);
}

// kinds 0 / 1: one thread per group of 4 consecutive elements (one draw); a chunk's ragged tail takes one draw per element
template <class T>
__global__ __launch_bounds__(256) void k_synth_fill(T* out, size_t n, const T* __restrict__ lut, uint64_t seed) {
	const size_t groups_per_chunk = CHUNK / 4;
	const size_t ngroups = (n + 3) / 4;
	for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
		const size_t c = g / groups_per_chunk, j = g % groups_per_chunk;
		const size_t a = c * CHUNK, b = a + CHUNK < n ? a + CHUNK : n; // the chunk's element range
		const size_t i = a + j * 4;
		if (i + 4 <= b) {
			const uint64_t r = draw(seed, c, j);
			out[i] = lut[r & 0xffff];
			out[i + 1] = lut[(r >> 16) & 0xffff];
			out[i + 2] = lut[(r >> 32) & 0xffff];
			out[i + 3] = lut[r >> 48];
		} else { // the last, partial group of the last chunk: draws nfull, nfull + 1, ... one element each
			const size_t nfull = (b - a) / 4;
			for (size_t e = i; e < b; ++e) {
				out[e] = lut[draw(seed, c, nfull + (e - i)) & 0xffff];
			}
		}
	}
}

// kind 2: one draw per gf4 word
__global__ __launch_bounds__(256) void k_synth_fill_gf4(uint32_t* out, size_t n, const uint8_t* __restrict__ lut, uint64_t seed) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const uint64_t r = draw(seed, i / CHUNK, i % CHUNK);
		uint32_t w = (uint32_t)(r >> 32) & 0xffffff00u;
		const uint32_t k = (uint32_t)(r & 7) * 3 + 8;
		w &= ~(7u << k);
		out[i] = w | lut[(r >> 8) & 0xffff];
	}
}

// binary32 -> fp8 e5m2 code, one round-to-nearest-even step, overflow to infinity: torch's float8_e5m2 conversion
__device__ __forceinline__ uint32_t f32_to_e5m2_rne(float f) {
	const uint32_t x = __float_as_uint(f);
	const uint32_t sign = (x >> 24) & 0x80u, absx = x & 0x7fffffffu;
	if (absx > 0x7f800000u) {
		return sign | 0x7fu;
	}
	if (absx >= 0x47700000u) { // >= 61440: rounds to infinity
		return sign | 0x7cu;
	}
	const int e = (int)(absx >> 23) - 127;
	if (e < -18) {
		return sign;
	}
	uint32_t man = (absx & 0x7fffffu) | 0x800000u, base = 0;
	int shift = 21 + (-14 - e);
	if (e >= -14) {
		shift = 21;
		base = (uint32_t)(e + 15) << 2;
		man &= 0x7fffffu;
	}
	uint32_t q = man >> shift;
	const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
	q += (rem > half || (rem == half && (q & 1))) ? 1 : 0;
	return sign | (base + q);
}

// one thread per group of 8 values (tools/convert.py:247-268)
__global__ __launch_bounds__(256) void k_quantize_gf4(const float* __restrict__ in, uint32_t* out, size_t ngroups) {
	for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
		float v[8];
		const float4 lo = ((const float4*)in)[2 * g], hi = ((const float4*)in)[2 * g + 1];
		v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
		int mi = 0; // first index of the largest magnitude (:252-253)
		for (int k = 1; k < 8; ++k) {
			mi = fabsf(v[k]) > fabsf(v[mi]) ? k : mi;
		}
		const uint32_t scode = f32_to_e5m2_rne(v[mi]); // :255
		const float s = __half2float(__ushort_as_half((unsigned short)(scode << 8)));
		uint32_t word = scode;
		for (int k = 0; k < 8; ++k) {
			float nrm = v[k] / s; // :257
			if (!(fabsf(nrm) <= 3.402823466e+38f)) { // nan, +-inf -> 0   (:258)
				nrm = 0.f;
			}
			// (x.half() * -4 + 4) evaluated in binary16, clamped to [0, 7], rounded half-to-even   (:261)
			const __half q16 = __hadd(__hmul(__float2half_rn(nrm), __float2half_rn(-4.0f)), __float2half_rn(4.0f));
			float q = __half2float(q16);
			q = q < 0.f ? 0.f : (q > 7.f ? 7.f : q);
			word += (uint32_t)(int)rintf(q) << (8 + 3 * k); // :263-264
		}
		out[g] = word;
	}
}

} // namespace

// kind 0: fp8 codes (uint8, lut of 65536 bytes)   1: fp16 patterns (uint16, lut of 65536 x 2 bytes)   2: gf4 words (uint32)
// `out` and `lut` are DEVICE pointers; runs on the null stream and returns when the fill is complete.
extern "C" void synth_fill_hip(void* out, size_t n, int kind, const void* lut, uint64_t seed) {
	const size_t threads = kind == 2 ? n : (n + 3) / 4;
	size_t blocks = (threads + 255) / 256;
	if (blocks > 256 * 64) {
		blocks = 256 * 64;
	}
	if (blocks == 0) {
		return;
	}
	if (kind == 0) {
		hipLaunchKernelGGL(k_synth_fill<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, 0, (uint8_t*)out, n, (const uint8_t*)lut, seed);
	} else if (kind == 1) {
		hipLaunchKernelGGL(k_synth_fill<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, 0, (uint16_t*)out, n, (const uint16_t*)lut, seed);
	} else {
		hipLaunchKernelGGL(k_synth_fill_gf4, dim3((unsigned)blocks), dim3(256), 0, 0, (uint32_t*)out, n, (const uint8_t*)lut, seed);
	}
	CK(hipGetLastError());
	CK(hipDeviceSynchronize());
}

// in: fp32 on the device, 8 * ngroups values; out: ngroups gf4 words on the device
extern "C" void quantize_gf4_hip(const float* in, uint32_t* out, size_t ngroups) {
	size_t blocks = (ngroups + 255) / 256;
	if (blocks > 256 * 64) {
		blocks = 256 * 64;
	}
	if (blocks == 0) {
		return;
	}
	hipLaunchKernelGGL(k_quantize_gf4, dim3((unsigned)blocks), dim3(256), 0, 0, in, out, ngroups);
	CK(hipGetLastError());
	CK(hipDeviceSynchronize());
}
