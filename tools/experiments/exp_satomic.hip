// Does gfx950 execute scalar memory atomics (s_atomic_add ... glc), and what do they cost under contention?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_satomic tools/experiments/exp_satomic.hip && /tmp/exp_satomic
// Every wave of a 512 x 256 grid takes tickets from one of `shards` counters (its block % shards) until the shard's pool is empty; every
// ticket must be handed out exactly once.  Prints correctness and the kernel's duration for a few shard counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned grab(unsigned* ctr) {
	unsigned v = 1;
	asm volatile("s_atomic_add %0, %1, 0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
	return v;
}
__device__ __forceinline__ unsigned peek(const unsigned* ctr) {
	unsigned v;
	asm volatile("s_load_dword %0, %1, 0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ctr) : "memory");
	return v;
}

__global__ void k_take(unsigned* ctr, unsigned* taken, int shards, int pool_per_shard, int use_peek) {
	const int shard = blockIdx.x % shards;
	unsigned* c = ctr + shard * 64; // 256 bytes apart
	for (;;) {
		if (use_peek && peek(c) >= (unsigned)pool_per_shard) {
			break;
		}
		const unsigned t = grab(c);
		if (t >= (unsigned)pool_per_shard) {
			break;
		}
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(&taken[shard * pool_per_shard + t], 1u);
		}
		// a little work per ticket: ~1 us of dependent math
		float x = (float)t;
		for (int i = 0; i < 600; ++i) {
			x = x * 1.0001f + 0.5f;
		}
		if (x == 12345.678f) {
			taken[0] = 7;
		}
	}
}

int main() {
	unsigned *ctr, *taken;
	const int total = 2048;
	CHECK(hipMalloc(&ctr, 8 * 64 * 4 * 4));
	CHECK(hipMalloc(&taken, total * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int use_peek = 0; use_peek < 2; ++use_peek) {
		for (int shards : {1, 8}) {
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipMemset(ctr, 0, 8 * 64 * 4 * 4));
				CHECK(hipMemset(taken, 0, total * 4));
				CHECK(hipEventRecord(e0));
				hipLaunchKernelGGL(k_take, dim3(512), dim3(256), 0, 0, ctr, taken, shards, total / shards, use_peek);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				std::vector<unsigned> h(total);
				CHECK(hipMemcpy(h.data(), taken, total * 4, hipMemcpyDeviceToHost));
				int bad = 0;
				for (unsigned v : h) {
					bad += v != 1;
				}
				printf("peek %d shards %d: %d tickets, %d not taken exactly once, %.1f us\n", use_peek, shards, total, bad, ms * 1e3);
			}
		}
	}
	return 0;
}
