// exp_overlap.hip -- EXPERIMENT (not product code): can the weight stream of kernel k+1 overlap the tail
// of kernel k?  Emulates one decode step as a chain of dependent streaming kernels (sizes of a
// Mistral-7B fp8 layer: 25.2 / 16.8 / 117.4 / 58.7 MB) in two modes:
//
//   mode 0  plain: one stream, kernel boundary between dependent kernels (what forward_hip does today)
//   mode 1  chained: kernels alternate between two streams; kernel k+1 issues its first weight loads,
//           then waits on a device-side completion counter of kernel k (agent-scope release/acquire per
//           cdna_hip_programming.md Guideline 16), then reads k's output vector.
//   mode 2  the same with system-scope loads/stores of the vector instead of the acquire fence
//   mode 3  ONE stream, dependent kernels launched with hipExtAnyOrderLaunch (AQL packet without the barrier
//           bit): the queue dispatches kernel k+1 as soon as kernel k's workgroups have all been dispatched and
//           slots free up, so k+1's prefetch overlaps k's tail; the data dependency is the same device-side
//           counter as in mode 1.  Kernel i keeps its barrier bit when i % depth == 0 (exp_set_depth), which
//           bounds how many spinning successors pile up behind a running kernel.  hip_ext.h says the flag is
//           "not supported on GFX9xx": exp_concurrency_anyorder() tests exactly that before any timing is read.
//   mode 4  mode 3 with system-scope loads/stores (as mode 2)
//
// Each kernel: 256 blocks x 512 threads; every wave streams 8 KiB tasks with 16 KiB in flight; the
// prologue reads the 16 KiB "activation" vector produced by the previous kernel (all blocks need all of
// it) and reduces it; the epilogue writes the wave's results into the next activation vector.
// The final activation vector must be bit-identical between the modes.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/experiments/libexp_overlap.so tools/experiments/exp_overlap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#define CK(x)                                                                                   \
	do {                                                                                        \
		hipError_t e_ = (x);                                                                    \
		if (e_ != hipSuccess) {                                                                 \
			fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			abort();                                                                            \
		}                                                                                       \
	} while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr16;

static int g_real = 0; // 1: every tile goes through the fp8 decode step's inner loop (tile_value<1>) in exp_chain / exp_deep
extern "C" void exp_set_real(int real) {
	g_real = real;
}

constexpr int BLOCK = 512;
constexpr int NW = BLOCK / 64;
constexpr int VEC = 4096; // floats in the activation vector

struct KArgs {
	const void* w;      // weights of this kernel
	size_t ntasks;      // 8 KiB tasks
	const float* xin;   // activation vector written by the previous kernel
	float* xout;        // activation vector this kernel writes (VEC floats; each task adds into one slot)
	unsigned* done_prev; // completion counter of the previous kernel (chained mode) or nullptr
	unsigned expect_prev;
	unsigned* done_me;
	unsigned* timeout; // set to 1 if a bounded spin gave up
	int chained;
};

// sum over the 64 lanes, VALU only (DPP row operations; the total lands in lane 63 and is broadcast with one v_readlane).
// (Round 2: the ds_bpermute butterfly this replaces cost ~0.3 us of dependent LDS round trips per task -- with one wave per
// SIMD that serial tail, not HBM, paced the persistent variants' catch-up after an edge.)
__device__ __forceinline__ float wave_sum(float v) {
#define EXP_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xf, true))
	v += EXP_DPP(v, 0xb1, 0xf);  // quad_perm [1,0,3,2]
	v += EXP_DPP(v, 0x4e, 0xf);  // quad_perm [2,3,0,1]
	v += EXP_DPP(v, 0x114, 0xf); // row_shr:4
	v += EXP_DPP(v, 0x118, 0xf); // row_shr:8
	v += EXP_DPP(v, 0x142, 0xa); // row_bcast:15
	v += EXP_DPP(v, 0x143, 0xc); // row_bcast:31
#undef EXP_DPP
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// What a wave does with one 8-KiB tile.  REAL = 0: a token xor-sum (the kernels are pure streaming: how fast can the
// launch structure possibly go).  REAL = 1: the fp8 decode step's inner loop -- every 16-byte lane-load is 16 e5m2 weights,
// converted in pairs (v_cvt_pk_f32_bf8) and multiplied into the staged vector read from LDS as float4s (v_pk_fma_f32), one
// VALU op and 4 LDS bytes per weight -- so that the consumers' duty cycle shows up in the measurement.
template <int REAL>
__device__ __forceinline__ float tile_value(const u32x4 (&t)[8], const float* xs, int lane) {
	if constexpr (REAL == 0) {
		unsigned acc = 0;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			acc += (t[u][0] ^ t[u][1]) + (t[u][2] ^ t[u][3]);
		}
		return (float)((acc & 0xff) + 1) * (1.0f / 4096.0f);
	} else {
		f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const f32x4 xv = ((const f32x4*)xs)[((u * 4 + i) * 64 + lane) & (VEC / 4 - 1)];
				a0 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_bf8((int)t[u][i], false), xv.lo, a0);
				a1 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_bf8((int)t[u][i], true), xv.hi, a1);
			}
		}
		return ((a0[0] + a0[1]) + (a1[0] + a1[1])) * (1.0f / 4096.0f) + 1.0f / 4096.0f;
	}
}

template <int REAL>
__global__ __launch_bounds__(BLOCK) void k_stream(KArgs a) {
	__shared__ float red[NW];
	__shared__ float xs[VEC];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const size_t W = (size_t)gridDim.x * NW;
	size_t t = (size_t)blockIdx.x * NW + wave;

	// 1. first two tasks' loads (16 KiB per wave) go out immediately
	u32x4 tile[2][8];
	size_t t0 = t < a.ntasks ? t : 0, t1 = t + W < a.ntasks ? t + W : 0;
#pragma unroll
	for (int u = 0; u < 8; ++u) {
		tile[0][u] = __builtin_nontemporal_load((gptr16)a.w + t0 * 512 + u * 64 + lane);
	}
#pragma unroll
	for (int u = 0; u < 8; ++u) {
		tile[1][u] = __builtin_nontemporal_load((gptr16)a.w + t1 * 512 + u * 64 + lane);
	}

	// 2. dependency on the previous kernel
	if (a.chained && a.done_prev) {
		if (threadIdx.x == 0) {
			unsigned spins = 0;
			while (__hip_atomic_load(a.done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.expect_prev) {
				__builtin_amdgcn_s_sleep(8);
				if (++spins > (1u << 16)) {
					*a.timeout = 1;
					break;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		}
		__syncthreads();
	}

	// 3. prologue: every block reads the whole activation vector, reduces it (emulates the norm)
	float4 xv[2];
	if (a.chained == 2) {
		// variant B: no acquire fence; the vector is read with system-scope (sc0 sc1) loads that bypass L1/L2
		const unsigned long long* p = (const unsigned long long*)a.xin;
		unsigned long long q0 = __hip_atomic_load(p + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		unsigned long long q1 = __hip_atomic_load(p + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		unsigned long long q2 = __hip_atomic_load(p + 2 * (threadIdx.x + BLOCK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		unsigned long long q3 = __hip_atomic_load(p + 2 * (threadIdx.x + BLOCK) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		xv[0] = make_float4(__uint_as_float((unsigned)q0), __uint_as_float((unsigned)(q0 >> 32)), __uint_as_float((unsigned)q1), __uint_as_float((unsigned)(q1 >> 32)));
		xv[1] = make_float4(__uint_as_float((unsigned)q2), __uint_as_float((unsigned)(q2 >> 32)), __uint_as_float((unsigned)q3), __uint_as_float((unsigned)(q3 >> 32)));
	} else {
		xv[0] = ((const float4*)a.xin)[threadIdx.x];
		xv[1] = ((const float4*)a.xin)[threadIdx.x + BLOCK];
	}
	float ss = xv[0].x * xv[0].x + xv[0].y * xv[0].y + xv[0].z * xv[0].z + xv[0].w * xv[0].w + xv[1].x * xv[1].x + xv[1].y * xv[1].y + xv[1].z * xv[1].z +
	           xv[1].w * xv[1].w;
	ss = wave_sum(ss);
	if (lane == 0) {
		red[wave] = ss;
	}
	((float4*)xs)[threadIdx.x] = xv[0];
	((float4*)xs)[threadIdx.x + BLOCK] = xv[1];
	__syncthreads();
	float tot = 0.f;
#pragma unroll
	for (int i = 0; i < NW; ++i) {
		tot += red[i];
	}
	const float scale = 1.0f / sqrtf(tot / VEC + 1e-5f);

	// 4. stream: two tasks always in flight
	for (;;) {
#pragma unroll
		for (int ph = 0; ph < 2; ++ph) {
			if (t >= a.ntasks) {
				goto finish;
			}
			const float tv = tile_value<REAL>(tile[ph], xs, lane);
			size_t t2 = t + 2 * W;
			size_t tl = t2 < a.ntasks ? t2 : 0;
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				tile[ph][u] = __builtin_nontemporal_load((gptr16)a.w + tl * 512 + u * 64 + lane);
			}
			float v = wave_sum(tv) * scale * xs[(t * 7) % VEC];
			if (lane == 0 && t < VEC) { // one writer per slot: the result is independent of timing
				float r = v + (float)(t % 13);
				if (a.chained == 2) {
					__hip_atomic_store(a.xout + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // sc0 sc1
				} else if (a.chained) {
					__hip_atomic_store(a.xout + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1 write-through
				} else {
					a.xout[t] = r;
				}
			}
			t += W;
		}
	}
finish:
	if (a.chained) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			__hip_atomic_fetch_add(a.done_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// mode 3: ONE persistent launch walks all phases; between phases a counter barrier whose latency is
// covered by the next phase's prefetched tiles (all blocks resident: grid <= CUs x blocks/CU).
struct Phase {
	const void* w;
	size_t ntasks;
};

__global__ __launch_bounds__(BLOCK) void k_persist(const Phase* ph, int nphases, float* x0, float* x1, unsigned* done, unsigned* timeout, int variant) {
	__shared__ float red[NW];
	__shared__ float xs[VEC];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const size_t W = (size_t)gridDim.x * NW;
	for (int p = 0; p < nphases; ++p) {
		const void* w = ph[p].w;
		const size_t ntasks = ph[p].ntasks;
		const float* xin = (p & 1) ? x1 : x0;
		float* xout = (p & 1) ? x0 : x1;
		size_t t = (size_t)blockIdx.x * NW + wave;
		u32x4 tile[2][8];
		size_t t0 = t < ntasks ? t : 0, t1 = t + W < ntasks ? t + W : 0;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[0][u] = __builtin_nontemporal_load((gptr16)w + t0 * 512 + u * 64 + lane);
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[1][u] = __builtin_nontemporal_load((gptr16)w + t1 * 512 + u * 64 + lane);
		}
		if (p > 0) {
			if (threadIdx.x == 0) {
				unsigned spins = 0;
				while (__hip_atomic_load(done + (p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
					__builtin_amdgcn_s_sleep(2);
					if (++spins > (1u << 18)) {
						*timeout = 1;
						break;
					}
				}
				if (variant == 0) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				}
			}
			__syncthreads();
		}
		float4 xv[2];
		if (variant == 1) {
			const unsigned long long* q = (const unsigned long long*)xin;
			unsigned long long q0 = __hip_atomic_load(q + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			unsigned long long q1 = __hip_atomic_load(q + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			unsigned long long q2 = __hip_atomic_load(q + 2 * (threadIdx.x + BLOCK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			unsigned long long q3 = __hip_atomic_load(q + 2 * (threadIdx.x + BLOCK) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			xv[0] = make_float4(__uint_as_float((unsigned)q0), __uint_as_float((unsigned)(q0 >> 32)), __uint_as_float((unsigned)q1), __uint_as_float((unsigned)(q1 >> 32)));
			xv[1] = make_float4(__uint_as_float((unsigned)q2), __uint_as_float((unsigned)(q2 >> 32)), __uint_as_float((unsigned)q3), __uint_as_float((unsigned)(q3 >> 32)));
		} else {
			xv[0] = ((const float4*)xin)[threadIdx.x];
			xv[1] = ((const float4*)xin)[threadIdx.x + BLOCK];
		}
		float ss = xv[0].x * xv[0].x + xv[0].y * xv[0].y + xv[0].z * xv[0].z + xv[0].w * xv[0].w + xv[1].x * xv[1].x + xv[1].y * xv[1].y +
		           xv[1].z * xv[1].z + xv[1].w * xv[1].w;
		ss = wave_sum(ss);
		__syncthreads(); // xs / red of the previous phase are no longer read
		if (lane == 0) {
			red[wave] = ss;
		}
		((float4*)xs)[threadIdx.x] = xv[0];
		((float4*)xs)[threadIdx.x + BLOCK] = xv[1];
		__syncthreads();
		float tot = 0.f;
#pragma unroll
		for (int i = 0; i < NW; ++i) {
			tot += red[i];
		}
		const float scale = 1.0f / sqrtf(tot / VEC + 1e-5f);
		bool go = true;
		while (go) {
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				if (t >= ntasks) {
					go = false;
					break;
				}
				unsigned acc = 0;
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					acc += (tile[h][u][0] ^ tile[h][u][1]) + (tile[h][u][2] ^ tile[h][u][3]);
				}
				size_t t2 = t + 2 * W;
				size_t tl = t2 < ntasks ? t2 : 0;
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					tile[h][u] = __builtin_nontemporal_load((gptr16)w + tl * 512 + u * 64 + lane);
				}
				float v = wave_sum((float)((acc & 0xff) + 1) * (1.0f / 4096.0f)) * scale * xs[(t * 7) % VEC];
				if (lane == 0 && t < VEC) {
					float r = v + (float)(t % 13);
					if (variant == 1) {
						__hip_atomic_store(xout + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					} else {
						__hip_atomic_store(xout + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
				t += W;
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			__hip_atomic_fetch_add(done + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

extern "C" double exp_persist(int variant, int n_layers, int iters, double* checksum, int grid) {
	static const size_t sizes[4] = {25165824, 16777216, 117440512, 58720256};
	const int NK = 4, total = n_layers * NK;
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	std::vector<void*> w(total);
	std::vector<Phase> hp(total);
	for (int i = 0; i < total; ++i) {
		CK(hipMalloc(&w[i], sizes[i % NK] + 65536));
		CK(hipMemset(w[i], 0x11 + i / NK + i % NK, sizes[i % NK] + 65536));
		hp[i].w = w[i];
		hp[i].ntasks = sizes[i % NK] / 8192;
	}
	Phase* dp;
	CK(hipMalloc(&dp, sizeof(Phase) * total));
	CK(hipMemcpy(dp, hp.data(), sizeof(Phase) * total, hipMemcpyHostToDevice));
	float* xbuf[2];
	CK(hipMalloc(&xbuf[0], VEC * 4 + 65536));
	CK(hipMalloc(&xbuf[1], VEC * 4 + 65536));
	std::vector<float> x0(VEC);
	for (int i = 0; i < VEC; ++i) {
		x0[i] = 0.001f * (i % 97) + 0.5f;
	}
	unsigned *done, *timeout;
	CK(hipMalloc(&done, 4 * (total + 1)));
	CK(hipMalloc(&timeout, 4));
	CK(hipMemset(timeout, 0, 4));
	auto run = [&]() {
		CK(hipMemcpyAsync(xbuf[0], x0.data(), VEC * 4, hipMemcpyHostToDevice, s));
		CK(hipMemsetAsync(done, 0, 4 * (total + 1), s));
		hipLaunchKernelGGL(k_persist, dim3(grid), dim3(BLOCK), 0, s, dp, total, xbuf[0], xbuf[1], done, timeout, variant);
	};
	run();
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, s));
	for (int i = 0; i < iters; ++i) {
		run();
	}
	CK(hipEventRecord(e1, s));
	CK(hipDeviceSynchronize());
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<float> xf(VEC);
	CK(hipMemcpy(xf.data(), xbuf[total & 1], VEC * 4, hipMemcpyDeviceToHost));
	double cs = 0;
	for (int i = 0; i < VEC; ++i) {
		cs += xf[i] * (1 + i % 5);
	}
	*checksum = cs;
	unsigned to = 0;
	CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
	if (to) {
		printf("  !! a bounded spin timed out (persistent variant %d)\n", variant);
	}
	for (void* p : w) {
		CK(hipFree(p));
	}
	CK(hipFree(dp));
	CK(hipFree(xbuf[0]));
	CK(hipFree(xbuf[1]));
	CK(hipFree(done));
	CK(hipFree(timeout));
	return (double)ms * 1e3 / ((double)iters * n_layers);
}

// mode "deep" (NOT YET RUN ON HARDWARE -- written at the end of round 1 for the first GPU minutes of round 2):
// the persistent launch again, but with the prefetch depth the arithmetic asks for.  A dependency edge inside a
// launch costs (barrier + re-read of the activation vector) MINUS what the waves had already asked HBM for; k_persist
// above holds 16 KiB per wave = 128 KiB per CU = 5.3 us of stream, a counter barrier plus the vector read is ~8 us, and
// the ~3 us difference is exactly a launch's fixed cost -- the measured tie (48.2 vs 49.9 us/layer).  The LDS ring of
// the guide's engine is the same 128 KiB (0.87x).  The register file is bigger than LDS: DEPTH tiles of 8 KiB per
// streaming wave (DEPTH = 6: 192 VGPRs, 7 waves: 336 KiB per CU = 14 us of stream) cover the whole edge.
// Two things make that depth usable:
//   * a wave with 48 loads in flight cannot poll global memory (its poll returns behind them) nor publish results
//     (vmcnt(0)): wave 0 of each workgroup is a COORDINATOR with an empty memory queue -- it stores the streamers'
//     results (handed over through LDS), arrives on / polls the grid counter, reloads and stages the activation vector
//     and raises an LDS flag; the 7 STREAMER waves touch global memory with weight loads only (counted vmcnt stays
//     exact) and wait on LDS;
//   * the streamers' load cursor runs ahead of their consume cursor ACROSS phase boundaries (weights depend on nothing).
// Same arithmetic per task as k_stream: the final vector must equal mode 0's bit for bit.
template <int K, int N, class F>
__device__ __forceinline__ bool static_steps(F& f) { // f(integral_constant<K>) for K = 0 .. N-1, stops when f returns true
	if constexpr (K < N) {
		if (f(std::integral_constant<int, K>())) {
			return true;
		}
		return static_steps<K + 1, N>(f);
	} else {
		return false;
	}
}

constexpr int DEEP_STREAMERS = NW - 1;
constexpr int DEEP_MAXT = 16; // tasks of one phase per streamer (14336 tasks / 1792 streamers = 8)
constexpr int DEEP_MAXP = 136; // phases per launch (a 32-layer token = 128 + classifier)

template <int DEPTH, int REAL>
__global__ __launch_bounds__(BLOCK) void k_deep(const Phase* __restrict__ ph_global, int nphases, float* x0, float* x1, unsigned* done, unsigned* timeout) {
	// the phase table lives in LDS: a streamer may touch global memory with weight loads ONLY (a table lookup compiled to a
	// vector load inside its cursor loops turns every counted s_waitcnt vmcnt(N) into vmcnt(0) -- seen in the first build)
	__shared__ Phase ph[DEEP_MAXP];
	for (int i = threadIdx.x; i < nphases; i += BLOCK) {
		ph[i] = ph_global[i];
	}
	__shared__ float xs[2][VEC];
	__shared__ float sc[2];                           // norm scale of the staged vector
	__shared__ float outv[2][DEEP_STREAMERS][DEEP_MAXT]; // results of a phase, per streamer
	__shared__ int outn[2][DEEP_STREAMERS];
	__shared__ int ready[2];    // phase number whose vector is staged in xs[p & 1] (+1), written by the coordinator
	__shared__ int finished[2]; // streamers that have deposited their results of phase p (slot p & 1)
	__shared__ int gave_up;     // a streamer's bounded spin timed out (streamers must not store to global memory: a store
	                            // behind a branch would make their counted vmcnt waits inexact); the coordinator reports it
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (threadIdx.x < 2) {
		ready[threadIdx.x] = 0;
		finished[threadIdx.x] = 0;
		gave_up = 0;
	}
	__syncthreads();
	const size_t GS = (size_t)gridDim.x * DEEP_STREAMERS;

	if (wave == 0) {
		// ---------------- coordinator: one wave, nothing in flight but what it is waiting for
		for (int p = 0; p <= nphases; ++p) {
			const float* xin = (p & 1) ? x1 : x0;
			if (p > 0) {
				// results of phase p-1 from this workgroup's streamers -> global (write-through), then the grid counter
				unsigned spins = 0;
				while (__hip_atomic_load(&finished[(p - 1) & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < DEEP_STREAMERS) {
					__builtin_amdgcn_s_sleep(1);
					if (++spins > (1u << 22)) {
						*timeout = 2;
						break;
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // the streamers' outn / outv
				float* xout = ((p - 1) & 1) ? x0 : x1;
				for (int i = lane; i < DEEP_STREAMERS * DEEP_MAXT; i += 64) {
					const int s = i / DEEP_MAXT, k = i % DEEP_MAXT;
					if (k < outn[(p - 1) & 1][s]) {
						const size_t t = (size_t)blockIdx.x * DEEP_STREAMERS + s + (size_t)k * GS;
						if (t < VEC) {
							__hip_atomic_store(xout + t, outv[(p - 1) & 1][s][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
					}
				}
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (lane == 0) {
					__hip_atomic_store(&finished[(p - 1) & 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					__hip_atomic_fetch_add(done + (p - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				if (p == nphases) {
					if (lane == 0 && gave_up) {
						*timeout = (unsigned)gave_up;
					}
					break;
				}
				spins = 0;
				while (__hip_atomic_load(done + (p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
					__builtin_amdgcn_s_sleep(2);
					if (++spins > (1u << 20)) {
						*timeout = 1;
						break;
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			}
			// stage the vector of phase p: 4096 floats by 64 lanes.  The sum of squares is taken in k_stream's order -- lane l
			// plays threads l, l + 64, ... of its 512-thread block, one butterfly per emulated wave, the eight wave totals
			// added in wave order -- so that the scale, and with it the final vector, equals mode 0's bit for bit.
			float tot = 0.f;
			for (int wv = 0; wv < NW; ++wv) {
				const int tid = wv * 64 + lane;
				const float4 a = ((const float4*)xin)[tid], b = ((const float4*)xin)[tid + BLOCK];
				((float4*)xs[p & 1])[tid] = a;
				((float4*)xs[p & 1])[tid + BLOCK] = b;
				float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
				tot += wave_sum(ss);
			}
			if (lane == 0) {
				sc[p & 1] = 1.0f / sqrtf(tot / VEC + 1e-5f);
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) {
				__hip_atomic_store(&ready[p & 1], p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		return;
	}

	// ---------------- streamers
	const int s = wave - 1;
	const size_t gs = (size_t)blockIdx.x * DEEP_STREAMERS + s;
	u32x4 tile[DEPTH][8];
	// cursors over the flattened (phase, task) sequence of this streamer
	int ip = 0, cp = 0; // issue / consume phase
	size_t it = gs, ct = gs;
	auto settle = [&](int& p, size_t& t) { // skip phases in which this streamer has no (more) task
		while (p < nphases && t >= ph[p].ntasks) {
			++p;
			t = gs;
		}
	};
	auto issue = [&](auto K) {
		constexpr int k = decltype(K)::value;
		settle(ip, it);
		const bool live = ip < nphases;
		const void* w = ph[live ? ip : nphases - 1].w; // past the end: re-read something valid, drop it
		const size_t t = live ? it : 0;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[k][u] = __builtin_nontemporal_load((gptr16)w + t * 512 + u * 64 + lane);
		}
		it += GS;
	};
	auto first = [&](auto K) {
		issue(K);
		return false;
	};
	static_steps<0, DEPTH>(first);
	int cur = -1, nout = 0; // phase whose vector this wave has seen ready; results deposited in it so far
	auto leave_phase = [&](int p) { // all tasks of phase p done: hand the count over
		if (lane == 0) {
			outn[p & 1][s] = nout;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__hip_atomic_fetch_add(&finished[p & 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		nout = 0;
	};
	auto step = [&](auto K) -> bool { // consume slot K, refill it; true when this streamer has no task left
		constexpr int k = decltype(K)::value;
		{
			// phases this streamer walks past (finished, or empty for it) are handed over in order
			while (cp < nphases && ct >= ph[cp].ntasks) {
				if (cur == cp) {
					leave_phase(cp);
				} else { // never entered: wait until its slot is free, then report zero results
					unsigned spins = 0;
					while (__hip_atomic_load(&ready[cp & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < cp + 1) {
						__builtin_amdgcn_s_sleep(1);
						if (++spins > (1u << 22)) {
							gave_up = 3;
							break;
						}
					}
					cur = cp;
					leave_phase(cp);
				}
				++cp;
				ct = gs;
			}
			if (cp >= nphases) {
				return true;
			}
			if (cur != cp) { // first task of a new phase: its vector must be staged
				unsigned spins = 0;
				while (__hip_atomic_load(&ready[cp & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < cp + 1) {
					__builtin_amdgcn_s_sleep(1);
					if (++spins > (1u << 22)) {
						gave_up = 4;
						break;
					}
				}
				cur = cp;
			}
			const float tv = tile_value<REAL>(tile[k], xs[cp & 1], lane);
			issue(K); // the slot is free again: DEPTH tasks ahead, whatever phase that is
			const float v = wave_sum(tv) * sc[cp & 1] * xs[cp & 1][(ct * 7) % VEC];
			if (lane == 0 && nout < DEEP_MAXT) {
				outv[cp & 1][s][nout] = v + (float)(ct % 13);
			}
			++nout;
			ct += GS;
		}
		return false;
	};
	while (!static_steps<0, DEPTH>(step)) {
	}
}

extern "C" double exp_deep(int depth, int n_layers, int iters, double* checksum) {
	static const size_t sizes[4] = {25165824, 16777216, 117440512, 58720256};
	const int NK = 4, total = n_layers * NK, grid = 256;
	if (total > DEEP_MAXP) {
		fprintf(stderr, "exp_deep: at most %d phases\n", DEEP_MAXP);
		return -1;
	}
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	std::vector<void*> w(total);
	std::vector<Phase> hp(total);
	for (int i = 0; i < total; ++i) {
		CK(hipMalloc(&w[i], sizes[i % NK] + 65536));
		CK(hipMemset(w[i], 0x11 + i / NK + i % NK, sizes[i % NK] + 65536));
		hp[i].w = w[i];
		hp[i].ntasks = sizes[i % NK] / 8192;
	}
	Phase* dp;
	CK(hipMalloc(&dp, sizeof(Phase) * total));
	CK(hipMemcpy(dp, hp.data(), sizeof(Phase) * total, hipMemcpyHostToDevice));
	float* xbuf[2];
	CK(hipMalloc(&xbuf[0], VEC * 4 + 65536));
	CK(hipMalloc(&xbuf[1], VEC * 4 + 65536));
	std::vector<float> x0(VEC);
	for (int i = 0; i < VEC; ++i) {
		x0[i] = 0.001f * (i % 97) + 0.5f;
	}
	unsigned *done, *timeout;
	CK(hipMalloc(&done, 4 * (total + 1)));
	CK(hipMalloc(&timeout, 4));
	CK(hipMemset(timeout, 0, 4));
	auto run = [&]() {
		CK(hipMemcpyAsync(xbuf[0], x0.data(), VEC * 4, hipMemcpyHostToDevice, s));
		CK(hipMemsetAsync(done, 0, 4 * (total + 1), s));
#define DEEP(d, r) hipLaunchKernelGGL((k_deep<d, r>), dim3(grid), dim3(BLOCK), 0, s, (const Phase*)dp, total, xbuf[0], xbuf[1], done, timeout)
		switch (depth * 2 + (g_real ? 1 : 0)) {
		case 4:
			DEEP(2, 0);
			break;
		case 5:
			DEEP(2, 1);
			break;
		case 8:
			DEEP(4, 0);
			break;
		case 9:
			DEEP(4, 1);
			break;
		case 10:
			DEEP(5, 0);
			break;
		case 11:
		case 13: // 6 tiles + the decode loop's registers spill (256 VGPRs at two waves per SIMD): 5 is the deepest REAL variant
			DEEP(5, 1);
			break;
		default:
			DEEP(6, 0);
		}
#undef DEEP
	};
	run();
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, s));
	for (int i = 0; i < iters; ++i) {
		run();
	}
	CK(hipEventRecord(e1, s));
	CK(hipDeviceSynchronize());
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<float> xf(VEC);
	CK(hipMemcpy(xf.data(), xbuf[total & 1], VEC * 4, hipMemcpyDeviceToHost));
	double cs = 0;
	for (int i = 0; i < VEC; ++i) {
		cs += xf[i] * (1 + i % 5);
	}
	*checksum = cs;
	unsigned to = 0;
	CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
	if (to) {
		printf("  !! a bounded spin timed out (deep, depth %d, code %u)\n", depth, to);
		fflush(stdout);
	}
	for (void* p : w) {
		CK(hipFree(p));
	}
	CK(hipFree(dp));
	CK(hipFree(xbuf[0]));
	CK(hipFree(xbuf[1]));
	CK(hipFree(done));
	CK(hipFree(timeout));
	return (double)ms * 1e3 / ((double)iters * n_layers);
}

// concurrency probe: do two kernels on two streams run at the same time?
__global__ void k_wait_flag(unsigned* flag, unsigned* seen, unsigned long long* cycles) {
	if (threadIdx.x == 0) {
		unsigned long long t0 = wall_clock64();
		unsigned ok = 0;
		for (unsigned i = 0; i < (1u << 20); ++i) {
			if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				ok = 1;
				break;
			}
			__builtin_amdgcn_s_sleep(16);
		}
		seen[blockIdx.x] = ok;
		cycles[blockIdx.x] = wall_clock64() - t0;
	}
}
__global__ void k_set_flag(unsigned* flag) {
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		__hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

extern "C" int exp_concurrency(int nblocks) {
	hipStream_t s0, s1;
	CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	unsigned *flag, *seen;
	unsigned long long* cyc;
	CK(hipMalloc(&flag, 4));
	CK(hipMalloc(&seen, 4 * nblocks));
	CK(hipMalloc(&cyc, 8 * nblocks));
	CK(hipMemset(flag, 0, 4));
	CK(hipMemset(seen, 0, 4 * nblocks));
	hipLaunchKernelGGL(k_wait_flag, dim3(nblocks), dim3(256), 0, s0, flag, seen, cyc);
	hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, s1, flag);
	CK(hipDeviceSynchronize());
	std::vector<unsigned> h(nblocks);
	std::vector<unsigned long long> c(nblocks);
	CK(hipMemcpy(h.data(), seen, 4 * nblocks, hipMemcpyDeviceToHost));
	CK(hipMemcpy(c.data(), cyc, 8 * nblocks, hipMemcpyDeviceToHost));
	int ok = 0;
	unsigned long long mx = 0;
	for (int i = 0; i < nblocks; ++i) {
		ok += h[i];
		mx = c[i] > mx ? c[i] : mx;
	}
	printf("concurrency probe: %d / %d waiting blocks saw the flag set by a kernel on another stream (max wait %llu ticks @100MHz)\n", ok, nblocks, mx);
	CK(hipFree(flag));
	CK(hipFree(seen));
	CK(hipFree(cyc));
	return ok;
}

static int g_depth = 0;
extern "C" void exp_set_depth(int depth) {
	g_depth = depth;
}

// the same probe on ONE stream: the waiting kernel first, then the flag-setting kernel launched with
// hipExtAnyOrderLaunch.  If the waiters see the flag, the second packet ran without waiting for the first to
// complete, i.e. the any-order flag is honoured on this chip; if not, they give up after their bounded spin.
extern "C" int exp_concurrency_anyorder(int nblocks) {
	hipStream_t s0;
	CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
	unsigned *flag, *seen;
	unsigned long long* cyc;
	CK(hipMalloc(&flag, 4));
	CK(hipMalloc(&seen, 4 * nblocks));
	CK(hipMalloc(&cyc, 8 * nblocks));
	CK(hipMemset(flag, 0, 4));
	CK(hipMemset(seen, 0, 4 * nblocks));
	CK(hipDeviceSynchronize());
	hipLaunchKernelGGL(k_wait_flag, dim3(nblocks), dim3(256), 0, s0, flag, seen, cyc);
	void* args[1] = {&flag};
	hipError_t e = hipExtLaunchKernel((const void*)k_set_flag, dim3(1), dim3(64), args, 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch);
	if (e != hipSuccess) {
		printf("any-order probe: hipExtLaunchKernel(hipExtAnyOrderLaunch) refused: %s\n", hipGetErrorString(e));
		(void)hipGetLastError();
		hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, s0, flag);
	}
	CK(hipDeviceSynchronize());
	std::vector<unsigned> h(nblocks);
	std::vector<unsigned long long> c(nblocks);
	CK(hipMemcpy(h.data(), seen, 4 * nblocks, hipMemcpyDeviceToHost));
	CK(hipMemcpy(c.data(), cyc, 8 * nblocks, hipMemcpyDeviceToHost));
	int ok = 0;
	unsigned long long mx = 0;
	for (int i = 0; i < nblocks; ++i) {
		ok += h[i];
		mx = c[i] > mx ? c[i] : mx;
	}
	printf("any-order probe: %d / %d waiting blocks saw the flag set by an any-order kernel queued BEHIND them on the same stream (max wait %llu ticks @100MHz)\n", ok,
	       nblocks, mx);
	fflush(stdout);
	CK(hipFree(flag));
	CK(hipFree(seen));
	CK(hipFree(cyc));
	return ok;
}

// One "token" = n_layers x 4 dependent streaming kernels.  Returns microseconds per layer; *checksum = sum of final vector.
extern "C" double exp_chain(int mode, int use_graph, int n_layers, int iters, double* checksum, int grid) {
	static const size_t sizes[4] = {25165824, 16777216, 117440512, 58720256};
	const int NK = 4;
	hipStream_t s[2];
	CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
	// weights: distinct buffers per layer (beyond the 256 MiB Infinity Cache in total)
	std::vector<void*> w(n_layers * NK);
	for (int l = 0; l < n_layers; ++l) {
		for (int k = 0; k < NK; ++k) {
			CK(hipMalloc(&w[l * NK + k], sizes[k] + 65536));
			CK(hipMemset(w[l * NK + k], 0x11 + l + k, sizes[k] + 65536));
		}
	}
	float* xbuf[2];
	CK(hipMalloc(&xbuf[0], VEC * 4 + 65536));
	CK(hipMalloc(&xbuf[1], VEC * 4 + 65536));
	std::vector<float> x0(VEC);
	for (int i = 0; i < VEC; ++i) {
		x0[i] = 0.001f * (i % 97) + 0.5f;
	}
	unsigned *done, *timeout;
	const int total = n_layers * NK;
	CK(hipMalloc(&done, 4 * (total + 1)));
	CK(hipMalloc(&timeout, 4));
	CK(hipMemset(timeout, 0, 4));

	auto enqueue = [&](bool capture_fork) {
		// x0 upload + counters reset happen on s[0] before everything
		CK(hipMemcpyAsync(xbuf[0], x0.data(), VEC * 4, hipMemcpyHostToDevice, s[0]));
		CK(hipMemsetAsync(done, 0, 4 * (total + 1), s[0]));
		hipEvent_t fork, join;
		if (mode == 1 || mode == 2) {
			CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
			CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
			CK(hipEventRecord(fork, s[0]));
			CK(hipStreamWaitEvent(s[1], fork, 0));
		}
		for (int i = 0; i < total; ++i) {
			KArgs a;
			a.w = w[i];
			a.ntasks = sizes[i % NK] / 8192;
			a.xin = xbuf[i & 1];
			a.xout = xbuf[(i + 1) & 1];
			a.done_prev = i > 0 ? done + (i - 1) : nullptr;
			a.expect_prev = grid;
			a.done_me = done + i;
			a.timeout = timeout;
			a.chained = mode >= 3 ? mode - 2 : mode;
			if (mode >= 3) {
				void* args[1] = {&a};
				const bool barrier = g_depth > 0 ? (i % g_depth == 0) : (i == 0);
				CK(hipExtLaunchKernel(g_real ? (const void*)k_stream<1> : (const void*)k_stream<0>, dim3(grid), dim3(BLOCK), args, 0, s[0], nullptr, nullptr,
				                      barrier ? 0 : hipExtAnyOrderLaunch));
			} else if (g_real) {
				hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(BLOCK), 0, s[mode >= 1 ? (i & 1) : 0], a);
			} else {
				hipLaunchKernelGGL(k_stream<0>, dim3(grid), dim3(BLOCK), 0, s[mode >= 1 ? (i & 1) : 0], a);
			}
		}
		if (mode == 1 || mode == 2) {
			CK(hipEventRecord(join, s[1]));
			CK(hipStreamWaitEvent(s[0], join, 0));
		}
	};

	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
	if (use_graph) {
		CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
		enqueue(true);
		CK(hipStreamEndCapture(s[0], &graph));
		CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
	}
	auto run = [&]() {
		if (use_graph) {
			CK(hipGraphLaunch(exec, s[0]));
		} else {
			enqueue(false);
		}
	};
	run();
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, s[0]));
	for (int i = 0; i < iters; ++i) {
		run();
	}
	CK(hipEventRecord(e1, s[0]));
	CK(hipDeviceSynchronize());
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<float> xf(VEC);
	CK(hipMemcpy(xf.data(), xbuf[total & 1], VEC * 4, hipMemcpyDeviceToHost));
	double cs = 0;
	for (int i = 0; i < VEC; ++i) {
		cs += xf[i] * (1 + i % 5);
	}
	*checksum = cs;
	unsigned to = 0;
	CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
	if (to) {
		printf("  !! a bounded spin timed out (mode %d)\n", mode);
	}
	for (void* p : w) {
		CK(hipFree(p));
	}
	CK(hipFree(xbuf[0]));
	CK(hipFree(xbuf[1]));
	CK(hipFree(done));
	CK(hipFree(timeout));
	return (double)ms * 1e3 / ((double)iters * n_layers);
}
