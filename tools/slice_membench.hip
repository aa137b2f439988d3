// slice_membench.hip -- what a BARE streaming-read kernel delivers at the sizes of the decode step's weight kernels (round 5).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/slice_membench tools/slice_membench.hip       run: tools/slice_membench
// tools/membench.py re-reads one buffer (sizes under 256 MiB are served by the Infinity Cache) and quotes large buffers; a decode kernel
// reads 17-131 MB of COLD weights once and ends.  Here every launch reads its own slice of a 3 GiB buffer (the slices cycle: a slice has
// been evicted from the Infinity Cache long before it is read again), launches back to back on one stream like the decode step's kernels,
// non-temporal 16-byte loads per lane, 2 x 256-thread workgroups per CU x 8 loads in flight per lane -- no activation vector, no LDS, no
// arithmetic beyond an XOR, no dependence between launches.  us per launch = (events around ITERS launches) / ITERS, boundary included:
// the yardstick for "T0 + bytes / rate" of a kernel of that size.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const u32x4* src, size_t n16, unsigned* sink) {
	unsigned acc = 0;
	const size_t stride = (size_t)gridDim.x * 256 * 8;
	for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n16; i += stride) {
		u32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const size_t j = i + (size_t)u * 256;
			v[u] = j < n16 ? __builtin_nontemporal_load(src + j) : (u32x4){0u, 0u, 0u, 0u};
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
		}
	}
	if (acc == 0x9e3779b9u) {
		*sink = acc; // never true in practice; keeps the loads alive
	}
}

// The same bytes in k_ffn_up's ACCESS PATTERN, still without any arithmetic or activation vector: 512 workgroups x 4 waves; a wave's tasks
// are hidden units j = wave, wave + 2048, ...; a task is row j of w1 and row j of w3 (ROW bytes each, the two matrices `half` bytes apart)
// walked in steps of TILE_KB KiB per row; a tile = that step of both rows; DEPTH tiles in flight per wave (the engine: 2 rows x 2 KiB, 2
// in flight).  Every load unconditional (clamped), like the engine's.
template <int TILE_KB, int DEPTH>
__global__ __launch_bounds__(256) void k_read_rows(const unsigned char* base, size_t half, int n_units, int row_bytes, unsigned* sink) {
	constexpr int LPT = 2 * TILE_KB; // wave-loads per tile (two rows)
	const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
	const int steps = row_bytes / (TILE_KB * 1024);
	const int my_units = (n_units - wave + nwaves - 1) / nwaves;
	const int ntiles = my_units * steps;
	auto load = [&](int t, u32x4(&v)[LPT]) {
		const int tc = t < ntiles ? t : ntiles - 1; // (clamped, never branched on)
		const int unit = wave + (tc / steps) * nwaves, st = tc % steps;
		const unsigned char* r0 = base + (size_t)unit * row_bytes + (size_t)st * TILE_KB * 1024 + lane * 16;
#pragma unroll
		for (int k = 0; k < TILE_KB; ++k) {
			v[2 * k] = __builtin_nontemporal_load((const u32x4*)(r0 + k * 1024));
			v[2 * k + 1] = __builtin_nontemporal_load((const u32x4*)(r0 + half + k * 1024));
		}
	};
	u32x4 ring[DEPTH][LPT];
	unsigned acc = 0;
#pragma unroll
	for (int d = 0; d < DEPTH; ++d) {
		load(d, ring[d]);
	}
	for (int t = 0; t < ntiles; t += DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
			for (int k = 0; k < LPT; ++k) {
				acc += ring[d][k][0] ^ ring[d][k][1] ^ ring[d][k][2] ^ ring[d][k][3];
			}
			load(t + d + DEPTH, ring[d]);
		}
	}
	if (acc == 0x9e3779b9u) {
		*sink = acc;
	}
}

int main() {
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	const size_t total = (size_t)3 << 30;
	unsigned char* buf;
	unsigned* sink;
	CK(hipMalloc(&buf, total));
	CK(hipMalloc(&sink, 4));
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	CK(hipMemsetAsync(buf, 0x5a, total, st));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const struct { const char* what; size_t bytes; } sizes[] = {
	    {"k_attn_out  (wo)        ", 16777216}, {"k_qkv       (wq|wk|wv)  ", 25165824}, {"k_ffn_down  (w2)        ", 58720256},
	    {"k_ffn_up    (w1 + w3)   ", 117440512}, {"k_output    (classifier)", 131072000}, {"(1 GiB, for scale)      ", (size_t)1 << 30}};
	printf("%s: %d CUs; bare non-temporal read kernel, every launch its own cold slice, back to back on one stream\n", prop.gcnArchName, ncu);
	printf("  size of                       MB      us / launch     GB/s     (of 8 TB/s)\n");
	for (const auto& s : sizes) {
		const size_t slice = (s.bytes + 4095) / 4096 * 4096, nslices = total / slice;
		const int iters = s.bytes > ((size_t)512 << 20) ? 12 : 200;
		for (int rep = 0; rep < 2; ++rep) { // (the first pass warms code and clocks; the second is quoted)
			CK(hipEventRecord(e0, st));
			for (int i = 0; i < iters; ++i) {
				hipLaunchKernelGGL(k_read, dim3(ncu * 2), dim3(256), 0, st, (const u32x4*)(buf + (size_t)(i % nslices) * slice), s.bytes / 16, sink);
			}
			CK(hipEventRecord(e1, st));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep == 1) {
				const double us = (double)ms * 1e3 / iters;
				printf("  %s  %8.2f    %8.2f     %7.0f     %.3f\n", s.what, s.bytes / 1e6, us, s.bytes / us / 1e3, s.bytes / us / 1e3 / 8000.0);
			}
		}
	}
	// k_ffn_up's access pattern on its own: Mistral-7B's 14336 hidden units x (4096 B of w1 + 4096 B of w3) = 117.44 MB per launch
	printf("\n  k_ffn_up's 117.44 MB in the row engine's access pattern (wave = hidden unit: its w1 row and its w3 row; no arithmetic, no vector):\n");
	const int n_units = 14336, row_bytes = 4096;
	const size_t half = (size_t)n_units * row_bytes, lslice = 2 * half, nls = total / lslice;
	auto pat = [&](const char* what, auto kern) {
		for (int rep = 0; rep < 2; ++rep) {
			CK(hipEventRecord(e0, st));
			for (int i = 0; i < 200; ++i) {
				hipLaunchKernelGGL(kern, dim3(ncu * 2), dim3(256), 0, st, (const unsigned char*)buf + (size_t)(i % nls) * lslice, half, n_units, row_bytes, sink);
			}
			CK(hipEventRecord(e1, st));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep == 1) {
				const double us = (double)ms * 1e3 / 200;
				printf("  %s  %8.2f us / launch   %7.0f GB/s\n", what, us, 2.0 * half / us / 1e3);
			}
		}
	};
	pat("tiles of 2 rows x 1 KiB, 2 in flight", k_read_rows<1, 2>);
	pat("tiles of 2 rows x 2 KiB, 2 in flight", k_read_rows<2, 2>); // the product's shape
	pat("tiles of 2 rows x 4 KiB, 2 in flight", k_read_rows<4, 2>);
	pat("tiles of 2 rows x 2 KiB, 4 in flight", k_read_rows<2, 4>);
	pat("tiles of 2 rows x 1 KiB, 4 in flight", k_read_rows<1, 4>);
	return 0;
}
