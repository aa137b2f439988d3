// exp_pfgemm.hip -- the prompt GEMM (calm_amd/csrc/prefill.hip.h: k_pf_gemm) on its own: correctness against a float64 dot
// product on sampled outputs, and the time per launch, for the shapes of a BASELINE layer.  EXPERIMENT TOOLING, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/experiments/exp_pfgemm tools/experiments/exp_pfgemm.hip && tools/experiments/exp_pfgemm [nb] [iters]
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

// S > 0: k_pf_gemm with S strips per wave;  S = -1: k_pf_gemm_wide;  S = -(10 * KS + 1): k_pf_gemm_wide with K split KS ways
template <int EPI, int S>
static void run(const char* name, int M, int K, int nb, int iters) {
	const int cols = (nb + 63) / 64;
	const size_t wbytes = (size_t)M * K;
	std::vector<uint8_t> W(wbytes * (EPI == PF_EPI_FFN_UP ? 2 : 1));
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
	const size_t fbytes = (size_t)cols * 64 * pf_steps(K) * 64 * sizeof(float);
	const size_t obytes = EPI == PF_EPI_FFN_UP ? (size_t)cols * 64 * pf_steps(M) * 64 * sizeof(float) : (size_t)nb * M * sizeof(float);
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
	a.xin = (const float4*)dXf, a.w0 = dW, a.w1 = dW + wbytes, a.K = K, a.M = M, a.nb = nb, a.out = dOut;
	a.clip = 3.4e38f;
	constexpr int UNITS = S > 0 ? PfTile<EPI, (S > 0 ? S : 1)>::UNITS : PfWide<EPI>::UNITS;
	a.ncols = cols;
	const int KS = S <= -10 ? (-S) / 10 : 1;
	a.ksplit = KS;
	const int ntile = 8 * (((M + UNITS - 1) / UNITS + 7) / 8) * cols;
	CK(hipMalloc(&a.partial, (size_t)ntile * KS * 16384 * sizeof(float)));
	CK(hipMalloc(&a.tile_count, (size_t)ntile * sizeof(unsigned)));
	CK(hipMemset(a.tile_count, 0, (size_t)ntile * sizeof(unsigned)));
	const dim3 grid = S > 0 ? dim3((M + UNITS - 1) / UNITS, cols) : dim3(pf_wide_grid((M + UNITS - 1) / UNITS, cols, KS));
	auto launch = [&]() {
		if constexpr (S > 0) {
			hipLaunchKernelGGL((k_pf_gemm<8, 16, EPI, (S > 0 ? S : 1)>), grid, dim3(256), 0, 0, a);
		} else {
			auto kern = k_pf_gemm_wide<8, 16, EPI, 1>;
			static bool once = false;
			if (!once) {
				CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PfWideA<8>::LDS_BYTES));
				once = true;
			}
			hipLaunchKernelGGL(kern, grid, dim3(256), PfWideA<8>::LDS_BYTES, 0, a);
		}
	};
	launch();
	CK(hipDeviceSynchronize());
	double worst = 0, scale = 0;
	if (EPI == PF_EPI_STORE) {
		std::vector<float> out((size_t)nb * M);
		CK(hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost));
		for (int smp = 0; smp < 512; ++smp) {
			const int t = (int)(rnd() % nb), u = smp < 8 ? (smp < 4 ? smp : M - 1 - (smp - 4)) : (int)(rnd() % M);
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
	const double flop = 2.0 * M * K * nb * (EPI == PF_EPI_FFN_UP ? 2 : 1);
	printf("%-10s M %6d K %6d nb %4d S %d grid %4u x %u : %8.1f us  %7.1f TFLOP/s (algorithmic)  %6.2f TB/s weights", name, M, K, nb, S, grid.x, grid.y, us,
	       flop / us * 1e-6, (double)W.size() / us * 1e-6);
	if (EPI == PF_EPI_STORE) {
		printf("   max |err| / max |y| = %.2e", worst / scale);
	}
	printf("\n");
	CK(hipFree(dW));
	CK(hipFree(dX));
	CK(hipFree(dXf));
	CK(hipFree(dOut));
	CK(hipFree(a.partial));
	CK(hipFree(a.tile_count));
}

int main(int argc, char** argv) {
	const int nb = argc > 1 ? atoi(argv[1]) : 256;
	const int iters = argc > 2 ? atoi(argv[2]) : 20;
	if (argc > 3) { // TinyLlama-1.1B's shapes (dim 2048, hidden 5632)
		run<PF_EPI_STORE, 2>("t-qkv", 2560, 2048, nb, iters);
		run<PF_EPI_STORE, 3>("t-qkv", 2560, 2048, nb, iters);
		run<PF_EPI_STORE, -1>("t-qkv", 2560, 2048, nb, iters);
		run<PF_EPI_STORE, -21>("t-qkv", 2560, 2048, nb, iters);
		run<PF_EPI_STORE, 2>("t-wo", 2048, 2048, nb, iters);
		run<PF_EPI_STORE, -1>("t-wo", 2048, 2048, nb, iters);
		run<PF_EPI_STORE, -21>("t-wo", 2048, 2048, nb, iters);
		run<PF_EPI_FFN_UP, 1>("t-ffn-up", 5632, 2048, nb, iters);
		run<PF_EPI_FFN_UP, -1>("t-ffn-up", 5632, 2048, nb, iters);
		run<PF_EPI_STORE, 2>("t-ffn-down", 2048, 5632, nb, iters);
		run<PF_EPI_STORE, -1>("t-ffn-down", 2048, 5632, nb, iters);
		run<PF_EPI_STORE, -21>("t-ffn-down", 2048, 5632, nb, iters);
		run<PF_EPI_STORE, -41>("t-ffn-down", 2048, 5632, nb, iters);
		return 0;
	}
	run<PF_EPI_STORE, 3>("qkv-like", 6144, 4096, nb, iters);
	run<PF_EPI_STORE, -1>("qkv-like", 6144, 4096, nb, iters);
	run<PF_EPI_STORE, -21>("qkv-like", 6144, 4096, nb, iters);
	run<PF_EPI_STORE, -41>("qkv-like", 6144, 4096, nb, iters);
	run<PF_EPI_STORE, 2>("wo-like", 4096, 4096, nb, iters);
	run<PF_EPI_STORE, -1>("wo-like", 4096, 4096, nb, iters);
	run<PF_EPI_STORE, -21>("wo-like", 4096, 4096, nb, iters);
	run<PF_EPI_STORE, -41>("wo-like", 4096, 4096, nb, iters);
	run<PF_EPI_STORE, -81>("wo-like", 4096, 4096, nb, iters);
	run<PF_EPI_FFN_UP, 1>("ffn-up", 14336, 4096, nb, iters);
	run<PF_EPI_FFN_UP, -1>("ffn-up", 14336, 4096, nb, iters);
	run<PF_EPI_FFN_UP, -21>("ffn-up", 14336, 4096, nb, iters);
	run<PF_EPI_STORE, 2>("ffn-down", 4096, 14336, nb, iters);
	run<PF_EPI_STORE, -1>("ffn-down", 4096, 14336, nb, iters);
	run<PF_EPI_STORE, -41>("ffn-down", 4096, 14336, nb, iters);
	run<PF_EPI_STORE, -81>("ffn-down", 4096, 14336, nb, iters);
	run<PF_EPI_STORE, 2>("ragged", 1000, 4128, nb < 200 ? nb : 200, iters);
	run<PF_EPI_STORE, -1>("ragged", 1000, 4128, nb < 200 ? nb : 200, iters);
	run<PF_EPI_STORE, -41>("ragged", 1000, 4128, nb < 200 ? nb : 200, iters);
	run<PF_EPI_STORE, 3>("classifier", 32000, 4096, nb, iters);
	run<PF_EPI_STORE, -1>("classifier", 32000, 4096, nb, iters);
	return 0;
}
