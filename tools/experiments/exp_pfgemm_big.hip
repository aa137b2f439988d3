// exp_pfgemm_big.hip -- a larger workgroup tile for the prompt GEMM, on its own: 512 units x 128
// tokens per 8-wave workgroup instead of k_pf_gemm_wide's 256 x 64 per 4 waves.  A wave owns 128 units x 64 tokens (4 x 2 accumulator
// tiles: 128 registers), the two waves of a unit strip share ONE LDS image of their 128 weight rows, the four waves of a token half
// share the B rows: 64 KB from the L2s per 64-column step for 4 x the multiply-adds of the 32 KB step of the wide kernel, and
// 24 ds_read_b128 + 8 ds_write_b128 per 64 MFMAs instead of 20 + 8 per 32.  Both operands are staged one step ahead into 2-slot rings
// (A 2 x 40 KB, B 2 x 32 KB: one workgroup per CU), one barrier per step.  fp8 weights, the plain store epilogue; correctness against
// a float64 dot product on sampled outputs, time per launch beside the product kernel's.  EXPERIMENT TOOLING, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/experiments/exp_pfgemm_big tools/experiments/exp_pfgemm_big.hip
//   tools/experiments/exp_pfgemm_big [tokens] [iters]
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "../../include/calm_abi.h"
#include "../../calm_amd/csrc/kernels.hip.h"
#include "../../calm_amd/csrc/prefill.hip.h"

using namespace calm;

#define CK(x)                                                                                       \
	do {                                                                                            \
		hipError_t e_ = (x);                                                                        \
		if (e_ != hipSuccess) {                                                                     \
			fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
			exit(1);                                                                                \
		}                                                                                           \
	} while (0)

constexpr int BIG_RS = 80;                  // A image: 64 bytes of a row per step + 16 (conflict-free ds_read_b128 for the 16-lane groups)
constexpr int BIG_A_STRIP = 128 * BIG_RS;   // one strip's 128 rows
constexpr int BIG_A_SLOT = 4 * BIG_A_STRIP; // 40 KB
constexpr int BIG_B_SLOT = 32 * 1024;       // 128 tokens x 64 columns x (hi, lo)
constexpr int BIG_LDS = 2 * (BIG_A_SLOT + BIG_B_SLOT);

template <int EPI, bool ROWS, bool PIPE>
__global__ __launch_bounds__(512, 1) void k_exp_gemm_big(PfGemmArgs a) {
	constexpr int NA = 4, NC = 2, P = 2;
	extern __shared__ u32x4 big_lds[];
	unsigned char* const lds = (unsigned char*)big_lds;
	u32x4(*bst)[32][64] = (u32x4(*)[32][64])lds; // [2][32][64]
	unsigned char* const abase = lds + 2 * BIG_B_SLOT;

	const int lane = lane_id(), wave = wave_id();
	const int j = lane & 31, kk = lane >> 5;
	const int strip = wave & 3, half = wave >> 2;
	const int ny = a.ncols; // 128-token columns
	const int idx = blockIdx.x >> 3;
	const int bx = (blockIdx.x & 7) + 8 * (idx / ny), by = idx % ny;
	if (bx * 512 >= a.M) {
		return;
	}
	const int unit_wg = bx * 512, tok_wg = by * 128;
	const size_t row_bytes = (size_t)a.K;
	const int row_pieces = (int)(row_bytes / 16);
	const int nsteps = pf_steps(a.K);

	const int apiece = lane & 3;
	const unsigned char* rowq[4];
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const int r = half * 64 + q * 16 + (lane >> 2);
		rowq[q] = (const unsigned char*)a.w0 + (size_t)min(unit_wg + strip * 128 + r, a.M - 1) * row_bytes;
	}
	const float4* xg = a.xin + (size_t)(tok_wg >> 5) * nsteps * 512 + lane;

	u32x4 fa[4], fb[4];
	auto load = [&](int sc) {
		const int scc = min(sc, nsteps - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int row = wave * 4 + r;
			fb[r] = *(const u32x4*)(xg + ((size_t)(row >> 3) * nsteps + scc) * 512 + (row & 7) * 64);
		}
		const int piece = min(scc * 4 + apiece, row_pieces - 1);
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			fa[q] = __builtin_nontemporal_load((gptr16)rowq[q] + piece);
		}
	};
	auto stage = [&](int slot, int sc) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			bst[slot][wave * 4 + r][lane] = fb[r];
		}
		unsigned char* img = abase + slot * BIG_A_SLOT + strip * BIG_A_STRIP;
		const bool valid = sc * 4 + apiece < row_pieces;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			*(u32x4*)(img + (half * 64 + q * 16 + (lane >> 2)) * BIG_RS + apiece * 16) = valid ? fa[q] : (u32x4){0u, 0u, 0u, 0u};
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
	// One step: multiply step s out of ring slot SLOT while step s + 1 (in registers since the step before) goes into the other slot
	// and step s + 2 is asked for -- two stores and two loads behind each of the four MFMA groups, so that the 64 KB of LDS stores of
	// a workgroup's step (~830 LDS cycles at the wide stores' rate) run in the shadow of the matrix cores instead of ahead of them
	// (PIPE; with PIPE = false: stage, fetch, then multiply, as k_pf_gemm_wide does).
	auto step = [&](auto SLOT, int s) {
		constexpr int slot = decltype(SLOT)::value;
		const unsigned char* img = abase + slot * BIG_A_SLOT + strip * BIG_A_STRIP;
		unsigned char* const nimg = abase + (slot ^ 1) * BIG_A_SLOT + strip * BIG_A_STRIP;
		const bool valid = (s + 1) * 4 + apiece < row_pieces;
		const int scc = min(s + 2, nsteps - 1);
		const int piece = min(scc * 4 + apiece, row_pieces - 1);
		auto move = [&](int q) { // register set q: step s + 1 -> LDS, step s + 2 -> registers
			bst[slot ^ 1][wave * 4 + q][lane] = fb[q];
			*(u32x4*)(nimg + (half * 64 + q * 16 + (lane >> 2)) * BIG_RS + apiece * 16) = valid ? fa[q] : (u32x4){0u, 0u, 0u, 0u};
			const int row = wave * 4 + q;
			fb[q] = *(const u32x4*)(xg + ((size_t)(row >> 3) * nsteps + scc) * 512 + (row & 7) * 64);
			fa[q] = __builtin_nontemporal_load((gptr16)rowq[q] + piece);
		};
		if constexpr (!PIPE) {
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				move(q);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		u32x4 w[NA][P];
#pragma unroll
		for (int n = 0; n < NA; ++n) {
#pragma unroll
			for (int i = 0; i < P; ++i) {
				w[n][i] = *(const u32x4*)(img + (32 * n + j) * BIG_RS + (kk * P + i) * 16);
			}
		}
		u32x4 bq[2][4];
		auto read_b = [&](u32x4(&q)[4], int m) {
#pragma unroll
			for (int c = 0; c < NC; ++c) {
#pragma unroll
				for (int hl = 0; hl < 2; ++hl) {
					q[c * 2 + hl] = bst[slot][(half * 2 + c) * 8 + m * 2 + hl][lane];
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
				wa[n] = pf_operand<8>(w[n][m / 2], m % 2);
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
			if constexpr (PIPE) {
				move(m);
			}
		}
		// A pieces + 4 reads | 4 reads, 16 MFMA, 2 stores, 2 loads | ... | 16 MFMA, 2 stores, 2 loads
		__builtin_amdgcn_sched_group_barrier(0x100, NA * P + 4, 0);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			if (m < 3) {
				__builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
			}
			__builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
			if constexpr (PIPE) {
				__builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
				__builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
			}
		}
		__syncthreads();
	};

	load(0);
	stage(0, 0);
	load(1);
	__syncthreads();
	for (int s = 0; s < nsteps; s += 2) {
		step(std::integral_constant<int, 0>(), s);
		if (s + 1 < nsteps) {
			step(std::integral_constant<int, 1>(), s + 1);
		}
	}
	const int unit0 = unit_wg + strip * 128, tok0 = tok_wg + half * 64;
	if constexpr (ROWS) {
		pf_epilogue_rows<16, EPI, NA>(a, acc, unit0, tok0, (float*)lds + wave * (32 * (32 * NA + 4)));
	} else {
		pf_epilogue<16, EPI, NA, NC>(a, acc, unit0, tok0, j, kk);
	}
}

__global__ void k_pack(void* out, const float* X, int K) { // row-major fp32 -> fragment-major hi / lo
	const int t = blockIdx.x;
	for (int i = threadIdx.x; i < K / 8; i += blockDim.x) {
		float v[8];
		for (int e = 0; e < 8; ++e) {
			v[e] = X[(size_t)t * K + 8 * i + e];
		}
		pf_store8(out, t, 8 * i, pf_steps(K), v);
	}
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return rng_state;
}
static float fp8_value(uint8_t b) {
	__half h = __ushort_as_half((unsigned short)(b << 8));
	return __half2float(h);
}

// form 0: k_pf_gemm_wide (the product kernel);  1: k_exp_gemm_big, stage / fetch / multiply in turn;  2: k_exp_gemm_big, staging behind the MFMA groups
static void run(const char* name, int form, int M, int K, int nb, int iters) {
	const int tok_alloc = (nb + 127) / 128 * 128;
	const size_t wbytes = (size_t)M * K;
	std::vector<uint8_t> W(wbytes);
	for (auto& b : W) {
		uint64_t r = rnd();
		b = (uint8_t)(((r & 1) << 7) | ((8 + (r >> 1) % 10) << 2) | ((r >> 8) & 3)); // 2^-7 .. 2^2, both signs
	}
	std::vector<float> X((size_t)nb * K);
	for (auto& x : X) {
		double u = (double)(rnd() >> 11) / 9007199254740992.0, v = (double)(rnd() >> 11) / 9007199254740992.0;
		x = (float)(sqrt(-2.0 * log(u + 1e-300)) * cos(6.283185307179586 * v));
	}
	uint8_t* dW;
	float *dX, *dOut;
	void* dXf;
	const size_t fbytes = (size_t)tok_alloc * pf_steps(K) * 64 * sizeof(float);
	const size_t obytes = (size_t)nb * M * sizeof(float);
	CK(hipMalloc(&dW, W.size() + 4096));
	CK(hipMalloc(&dX, X.size() * 4));
	CK(hipMalloc(&dXf, fbytes));
	CK(hipMalloc(&dOut, obytes));
	CK(hipMemset(dXf, 0, fbytes));
	CK(hipMemset(dOut, 0, obytes));
	CK(hipMemcpy(dW, W.data(), W.size(), hipMemcpyHostToDevice));
	CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_pack, dim3(nb), dim3(256), 0, 0, dXf, dX, K);
	PfGemmArgs a;
	memset(&a, 0, sizeof(a));
	a.xin = (const float4*)dXf, a.w0 = dW, a.w1 = dW, a.K = K, a.M = M, a.nb = nb, a.out = dOut;
	a.clip = 3.4e38f;
	a.ksplit = 1;
	const int units = form == 0 ? 256 : 512, tcol = form == 0 ? 64 : 128;
	a.ncols = (nb + tcol - 1) / tcol;
	const dim3 grid(pf_wide_grid((M + units - 1) / units, a.ncols));
	auto launch = [&]() {
		if (form == 0) {
			auto kern = k_pf_gemm_wide<8, 16, PF_EPI_STORE, 1>;
			CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PfWideA<8>::LDS_BYTES));
			hipLaunchKernelGGL(kern, grid, dim3(256), PfWideA<8>::LDS_BYTES, 0, a);
		} else if (form == 1) {
			auto kern = k_exp_gemm_big<PF_EPI_STORE, true, false>;
			CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS));
			hipLaunchKernelGGL(kern, grid, dim3(512), BIG_LDS, 0, a);
		} else if (form == 2) {
			auto kern = k_exp_gemm_big<PF_EPI_STORE, true, true>;
			CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS));
			hipLaunchKernelGGL(kern, grid, dim3(512), BIG_LDS, 0, a);
		} else { // the product's form of it (calm_amd/csrc/prefill.hip.h)
			auto kern = k_pf_gemm_big<8, PF_EPI_STORE>;
			CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PfBigA<8>::LDS_BYTES));
			hipLaunchKernelGGL(kern, grid, dim3(512), PfBigA<8>::LDS_BYTES, 0, a);
		}
	};
	launch();
	CK(hipDeviceSynchronize());
	double worst = 0, scale = 0;
	{
		std::vector<float> out((size_t)nb * M);
		CK(hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost));
		for (int smp = 0; smp < 768; ++smp) {
			const int t = smp < 16 ? (smp < 8 ? smp * 17 % nb : nb - 1 - (smp - 8)) : (int)(rnd() % nb);
			const int u = smp < 8 ? (smp < 4 ? smp : M - 1 - (smp - 4)) : (int)(rnd() % M);
			double ref = 0;
			for (int k = 0; k < K; ++k) {
				ref += (double)fp8_value(W[(size_t)u * K + k]) * (double)X[(size_t)t * K + k];
			}
			worst = fmax(worst, fabs(ref - out[(size_t)t * M + u]));
			scale = fmax(scale, fabs(ref));
		}
	}
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	for (int i = 0; i < 3; ++i) {
		launch();
	}
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < iters; ++i) {
		launch();
	}
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / iters;
	const double flop = 2.0 * M * K * nb;
	printf("%-10s %-22s M %6d K %6d tokens %4d grid %4u : %8.1f us  %7.1f TFLOP/s (algorithmic)   max |err| / max |y| = %.2e\n", name,
	       form == 0 ? "wide 256 x 64 (product)" : (form == 1 ? "big 512 x 128" : (form == 2 ? "big 512 x 128 pipelined" : "k_pf_gemm_big (product)")), M, K, nb, grid.x, us, flop / us * 1e-6, worst / scale);
	CK(hipFree(dW));
	CK(hipFree(dX));
	CK(hipFree(dXf));
	CK(hipFree(dOut));
}

int main(int argc, char** argv) {
	const int nb = argc > 1 ? atoi(argv[1]) : 1024;
	const int iters = argc > 2 ? atoi(argv[2]) : 20;
	const char* only = argc > 3 ? argv[3] : nullptr; // one shape (for counter passes)
	const int lo = argc > 4 ? atoi(argv[4]) : 0, hi = argc > 4 ? atoi(argv[4]) + 1 : 4;
	struct Shape {
		const char* name;
		int M, K, cap;
	} shapes[] = {{"ffn-up-like", 28672, 4096, 1 << 30}, {"classifier", 32000, 4096, 1 << 30}, {"qkv-like", 6144, 4096, 1 << 30}, {"ffn-down", 4096, 14336, 1 << 30}, {"wo-like", 4096, 4096, 1 << 30},
	              {"ragged", 1000, 4128, 200}};
	for (const Shape& sh : shapes) {
		if (only && strcmp(only, sh.name)) {
			continue;
		}
		for (int form = lo; form < hi; ++form) {
			run(sh.name, form, sh.M, sh.K, nb < sh.cap ? nb : sh.cap, iters);
		}
	}
	return 0;
}
