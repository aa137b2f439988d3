// exp_engine.hip -- EXPERIMENT (not product code): the one persistent structure the round-2 measurements do not rule out.
//
// k_burst (exp_mega.hip) showed why a register ring cannot bank an edge: a CU accepts ~64 wave-loads, the issuing wave blocks
// beyond that, and in steady state the ring is full of landed tiles.  Data must LAND somewhere other than the issuing wave's
// registers, and the buffer must start every edge empty.  That is the loader/consumer engine of the CDNA guide:
//   * wave 0 of every workgroup (one per CU) is a LOADER: it walks the workgroup's tasks across ALL phases and streams each
//     8 KiB tile into the next slot of a 16-slot LDS ring with LDS-DMA (global_load_lds_dwordx4, non-temporal), blocked only by
//     the CU's memory queue and by a full ring; it publishes "slot g has landed" through a counted s_waitcnt vmcnt;
//   * NC CONSUMER waves read tiles from LDS (they are faster than HBM, so the ring is nearly empty in steady state), never touch
//     global memory except at an edge -- where the last consumer out publishes the workgroup's results (write-through burst, one
//     arrival on a per-XCD counter shard), consumer 0 polls the shards, and all consumers reload and stage the next vector --
//     while the loader keeps filling the ring: up to 128 KiB = 5.2 us of the CU's share of the stream per edge.
// Same arithmetic per task as k_stream / k_mega: the final vector must equal exp_chain's bit for bit.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/experiments/libexp_engine.so tools/experiments/exp_engine.hip
#include "exp_overlap.hip"

constexpr int EG_MAXP = 136;
__device__ int eg_noedge; // diagnostic: 1 = the grid-wide wait at an edge is skipped (wrong results, in-phase rate only)
constexpr int EG_MAXT = 64;  // tasks of one phase per workgroup
constexpr int EG_SHARD = 32; // uints between counter shards (128 bytes)

struct EPhase {
	const void* w;
	unsigned ntasks;
	unsigned K; // tasks per workgroup (ceil)
};

template <int SYS>
__device__ __forceinline__ f32x4 eg_load16(const float* p) {
	f32x4 v;
	if constexpr (SYS) {
		asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	} else {
		asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	}
	return v;
}

template <int N>
__device__ __forceinline__ void eg_wait_vmcnt() {
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS accesses the compiler must not see (it orders every LDS access it knows about behind pending LDS-DMA with vmcnt(0),
// which would drain the loader's stream at every flag read or write)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
	return (unsigned)(size_t)(lds_ptr_t)p;
}
__device__ __forceinline__ int lds_read_i32(unsigned addr) {
	int v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return v;
}
__device__ __forceinline__ unsigned long long lds_read_u64(unsigned addr) {
	unsigned long long v;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return v;
}
__device__ __forceinline__ void lds_write_i32(unsigned addr, int v) {
	asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// The loader wave: its own function with its own (small) register allocation -- inlined into the kernel it shared the consumers'
// pressure, its lane id was spilled to scratch and every reload drained the DMA queue (s_waitcnt vmcnt(0)).
template <int NC, int R, int D, int NL>
__device__ __forceinline__ void eg_loader(const EPhase* ph, const int* cum, unsigned char* ring, int* landed, int* done, int* gave_up, int nphases, int G, int lane, int lw) {
	const unsigned a_landed = lds_addr(landed), a_done = lds_addr(done), a_cum = lds_addr(cum), a_ph = lds_addr(ph), a_ring = lds_addr(ring);
	int p = -1, p_end = 0, p_begin = 0; // tasks [p_begin, p_end) of this workgroup belong to phase p
	const unsigned char* wbase = nullptr;
	unsigned K = 0;
	int n = 0; // tiles this loader has issued
	for (int g = lw; g < G; g += NL, ++n) {
		while (g >= p_end) { // next phase with work: its descriptor comes from LDS once
			++p;
			p_begin = lds_read_i32(a_cum + 4 * p);
			p_end = lds_read_i32(a_cum + 4 * (p + 1));
			wbase = (const unsigned char*)lds_read_u64(a_ph + 16 * p);
			K = (unsigned)lds_read_i32(a_ph + 16 * p + 12);
		}
		const unsigned t = blockIdx.x * K + (unsigned)(g - p_begin);
		if (g >= R) { // the slot's previous tenant (task g - R, consumer (g - R) % NC, its ((g - R) / NC)-th) must be consumed
			const int c = (g - R) % NC, j = (g - R) / NC;
			unsigned spins = 0;
			while (lds_read_i32(a_done + 4 * c) <= j) {
				__builtin_amdgcn_s_sleep(1);
				if (++spins > (1u << 22)) {
					lds_write_i32(lds_addr(gave_up), 5);
					break;
				}
			}
		}
		const unsigned char* src = wbase + (size_t)t * 8192 + lane * 16;
		const unsigned slot = a_ring + (unsigned)(g % R) * 8192;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * 1024), (lds_ptr_t)(size_t)(slot + u * 1024), 16, 0, 2 /* nt */);
		}
		if (n + 1 >= D) { // all but this loader's latest D - 1 tiles have landed
			eg_wait_vmcnt<8 * (D - 1)>();
			if (lane == 0) {
				lds_write_i32(a_landed, n + 2 - D);
			}
		}
	}
	eg_wait_vmcnt<0>();
	if (lane == 0) {
		lds_write_i32(a_landed, n);
	}
}

// NC consumers, ring of R slots of 8 KiB, at most D tiles (8 DMA instructions each) in flight before the loader looks at the
// oldest; REAL: the fp8 decode inner loop per tile; SYS: the vector travels with system-scope accesses instead of sc1
template <int NC, int R, int D, int REAL, int SYS, int NL>
__global__ __launch_bounds__((NC + NL) * 64) void k_engine(const EPhase* __restrict__ ph_global, int nphases, float* x0, float* x1, unsigned* cnt, unsigned* timeout) {
	__shared__ __attribute__((aligned(1024))) unsigned char ring[R][8192];
	__shared__ __attribute__((aligned(16))) float xs[VEC];
	__shared__ float red[NW];
	__shared__ float outv[EG_MAXT];
	__shared__ EPhase ph[EG_MAXP];
	__shared__ int cum[EG_MAXP + 1]; // this workgroup's tasks before phase p
	__shared__ int landed[4];        // tiles landed in the ring so far, per loader wave (monotonic): loader l owns tiles g = l, l + NL, ...
	__shared__ int done[NC];         // tasks finished by each consumer (monotonic)
	__shared__ int closed;           // consumers x phases they have left (monotonic): the NC-th of a phase closes it
	__shared__ int gate;             // phases whose grid-wide arrival consumer 0 has seen (monotonic)
	__shared__ int staged;           // consumer x phases staged (monotonic)
	__shared__ int gave_up;
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	for (int i = threadIdx.x; i < nphases; i += (NC + NL) * 64) {
		ph[i] = ph_global[i];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int c = 0;
		for (int p = 0; p < nphases; ++p) {
			cum[p] = c;
			const long base = (long)blockIdx.x * ph[p].K;
			long n = (long)ph[p].ntasks - base;
			n = n < 0 ? 0 : (n > (long)ph[p].K ? (long)ph[p].K : n);
			c += (int)n;
		}
		cum[nphases] = c;
		landed[0] = landed[1] = landed[2] = landed[3] = 0, closed = 0, gate = 0, staged = 0, gave_up = 0;
		for (int i = 0; i < NC; ++i) {
			done[i] = 0;
		}
	}
	__syncthreads();
	const int G = cum[nphases]; // this workgroup's tasks over the whole launch

	if (wave < NL) {
		eg_loader<NC, R, D, NL>(ph, cum, &ring[0][0], &landed[wave], done, &gave_up, nphases, G, lane, wave);
		return;
	}

	// ---------------------------------------------------------------------- consumers
	const int c = wave - NL;
	const ptrdiff_t xstep = x1 - x0;
	auto wait_at_least = [&](int* what, int target, int code) __attribute__((always_inline)) {
		unsigned spins = 0;
		while (__hip_atomic_load(what, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > (1u << 22)) {
				gave_up = code;
				break;
			}
		}
	};
	// stage the vector of phase p (every consumer its share; sum of squares in k_stream's order, see exp_mega.hip)
	auto stage_vector = [&](int p) __attribute__((always_inline)) {
		const float* xin = x0 + (ptrdiff_t)(p & 1) * xstep; // (a select between the two kernel arguments becomes a scratch lookup table)
		constexpr int RR = (NW + NC - 1) / NC;
		f32x4 a[RR], b[RR];
#pragma unroll
		for (int r = 0; r < RR; ++r) {
			const int wv = c + r * NC;
			const int tid = (wv < NW ? wv : 0) * 64 + lane;
			a[r] = eg_load16<SYS>(xin + 4 * tid);
			b[r] = eg_load16<SYS>(xin + 4 * (tid + BLOCK));
		}
#pragma unroll
		for (int r = 0; r < RR; ++r) {
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[r]), "+v"(b[r])::"memory");
		}
#pragma unroll
		for (int r = 0; r < RR; ++r) {
			const int wv = c + r * NC;
			if (wv < NW) {
				const int tid = wv * 64 + lane;
				((f32x4*)xs)[tid] = a[r];
				((f32x4*)xs)[tid + BLOCK] = b[r];
				float ss = a[r].x * a[r].x + a[r].y * a[r].y + a[r].z * a[r].z + a[r].w * a[r].w + b[r].x * b[r].x + b[r].y * b[r].y + b[r].z * b[r].z + b[r].w * b[r].w;
				ss = wave_sum(ss);
				if (lane == 0) {
					red[wv] = ss;
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0) {
			__hip_atomic_fetch_add(&staged, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	};
	// this consumer leaves phase p: the last of the NC publishes the workgroup's results and arrives; then everybody gets
	// phase p + 1's vector staged
	auto leave_phase = [&](int p) __attribute__((always_inline)) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		int before = 0;
		if (lane == 0) {
			before = __hip_atomic_fetch_add(&closed, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		before = __builtin_amdgcn_readfirstlane(before);
		if (before == NC * (p + 1) - 1) { // the closer
			float* xout = x1 - (ptrdiff_t)(p & 1) * xstep;
			const unsigned nt = ph[p].ntasks, base = blockIdx.x * ph[p].K;
			const unsigned mine = (unsigned)(cum[p + 1] - cum[p]);
			for (unsigned i = lane; i < mine; i += 64) {
				const unsigned t = base + i;
				if (t < nt && t < VEC) {
					if constexpr (SYS) {
						__hip_atomic_store(xout + t, outv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					} else {
						__hip_atomic_store(xout + t, outv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (lane == 0) {
				__hip_atomic_fetch_add(cnt + (blockIdx.x & 7) * EG_SHARD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		if (p + 1 >= nphases) {
			return;
		}
		if (c == 0) { // the poller
			const unsigned target = eg_noedge ? 0u : (gridDim.x / 8) * (unsigned)(p + 1);
			unsigned spins = 0;
			for (;;) {
				unsigned v = target;
				if (lane < 8) {
					v = __hip_atomic_load(cnt + lane * EG_SHARD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				if (__builtin_amdgcn_ballot_w64(v < target) == 0) {
					break;
				}
				__builtin_amdgcn_s_sleep(1);
				if (++spins > (1u << 20)) {
					*timeout = 1;
					break;
				}
			}
			if (lane == 0) {
				__hip_atomic_store(&gate, p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else {
			wait_at_least(&gate, p + 1, 6);
		}
		stage_vector(p + 1);
	};

	stage_vector(0);
	int p = 0;
	float scale = 0.f;
	int scaled_for = -1;
	for (int g = c; g < G; g += NC) {
		while (g >= cum[p + 1]) {
			leave_phase(p);
			++p;
		}
		if (scaled_for != p) {
			wait_at_least(&staged, NC * (p + 1), 7);
			float tot = 0.f;
#pragma unroll
			for (int i = 0; i < NW; ++i) {
				tot += red[i];
			}
			scale = 1.0f / sqrtf(tot / VEC + 1e-5f);
			scaled_for = p;
		}
		wait_at_least(&landed[g % NL], g / NL + 1, 8);
		const unsigned k = (unsigned)(g - cum[p]);
		const unsigned t = blockIdx.x * ph[p].K + k;
		u32x4 tile[8];
		const u32x4* slot = (const u32x4*)&ring[g % R][0];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[u] = slot[u * 64 + lane];
		}
		const float tv = tile_value<REAL>(tile, xs, lane);
		const float v = wave_sum(tv) * scale * xs[(t * 7) % VEC];
		if (lane == 0) {
			outv[k] = v + (float)(t % 13);
			__hip_atomic_store(&done[c], g / NC + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // (the tile's reads have returned: tv depends on them)
		}
	}
	while (p < nphases) { // phases after this consumer's last task
		leave_phase(p);
		++p;
	}
	if (c == 0 && lane == 0 && gave_up) {
		*timeout = (unsigned)gave_up;
	}
}

template <int NC, int R, int D, int REAL, int SYS, int NL>
static double engine_run(int n_layers, int iters, double* checksum) {
	static const size_t sizes[4] = {25165824, 16777216, 117440512, 58720256};
	const int NK = 4, total = n_layers * NK, grid = 256;
	if (total > EG_MAXP) {
		fprintf(stderr, "exp_engine: at most %d phases\n", EG_MAXP);
		return -1;
	}
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	std::vector<void*> w(total);
	std::vector<EPhase> hp(total);
	for (int i = 0; i < total; ++i) {
		CK(hipMalloc(&w[i], sizes[i % NK] + 65536));
		CK(hipMemset(w[i], 0x11 + i / NK + i % NK, sizes[i % NK] + 65536));
		hp[i].w = w[i];
		hp[i].ntasks = (unsigned)(sizes[i % NK] / 8192);
		hp[i].K = (hp[i].ntasks + grid - 1) / grid;
		if (hp[i].K > EG_MAXT) {
			fprintf(stderr, "exp_engine: %u tasks per workgroup exceed %d\n", hp[i].K, EG_MAXT);
			return -1;
		}
	}
	EPhase* dp;
	CK(hipMalloc(&dp, sizeof(EPhase) * total));
	CK(hipMemcpy(dp, hp.data(), sizeof(EPhase) * total, hipMemcpyHostToDevice));
	float* xbuf[2];
	CK(hipMalloc(&xbuf[0], VEC * 4 + 65536));
	CK(hipMalloc(&xbuf[1], VEC * 4 + 65536));
	std::vector<float> x0(VEC);
	for (int i = 0; i < VEC; ++i) {
		x0[i] = 0.001f * (i % 97) + 0.5f;
	}
	unsigned *cnt, *timeout;
	CK(hipMalloc(&cnt, 4 * EG_SHARD * 8));
	CK(hipMalloc(&timeout, 4));
	CK(hipMemset(timeout, 0, 4));
	auto kern = k_engine<NC, R, D, REAL, SYS, NL>;
	auto run = [&]() {
		CK(hipMemcpyAsync(xbuf[0], x0.data(), VEC * 4, hipMemcpyHostToDevice, s));
		CK(hipMemsetAsync(cnt, 0, 4 * EG_SHARD * 8, s));
		hipLaunchKernelGGL(kern, dim3(grid), dim3((NC + NL) * 64), 0, s, (const EPhase*)dp, total, xbuf[0], xbuf[1], cnt, timeout);
	};
	run();
	CK(hipGetLastError());
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
		printf("  !! a bounded spin timed out (engine NC=%d R=%d D=%d NL=%d, code %u)\n", NC, R, D, NL, to);
	}
	fflush(stdout);
	for (void* p : w) {
		CK(hipFree(p));
	}
	CK(hipFree(dp));
	CK(hipFree(xbuf[0]));
	CK(hipFree(xbuf[1]));
	CK(hipFree(cnt));
	CK(hipFree(timeout));
	CK(hipStreamDestroy(s));
	return (double)ms * 1e3 / ((double)iters * n_layers);
}

// config = NL * 100000 + NC * 1000 + R * 10 + D;  flags: bit 0 REAL, bit 1 system-scope vector
extern "C" void exp_engine_noedge(int v) {
	CK(hipMemcpyToSymbol(HIP_SYMBOL(eg_noedge), &v, sizeof(v)));
}

extern "C" double exp_engine(int config, int flags, int n_layers, int iters, double* checksum) {
#define EG(nl, nc, r, d)                                                        \
	if (config == nl * 100000 + nc * 1000 + r * 10 + d) {                       \
		switch (flags & 3) {                                                    \
		case 0:                                                                 \
			return engine_run<nc, r, d, 0, 0, nl>(n_layers, iters, checksum);   \
		case 1:                                                                 \
			return engine_run<nc, r, d, 1, 0, nl>(n_layers, iters, checksum);   \
		case 2:                                                                 \
			return engine_run<nc, r, d, 0, 1, nl>(n_layers, iters, checksum);   \
		default:                                                                \
			return engine_run<nc, r, d, 1, 1, nl>(n_layers, iters, checksum);   \
		}                                                                       \
	}
	EG(1, 4, 16, 6)
	EG(1, 7, 16, 6)
	EG(1, 4, 8, 4)
	EG(2, 4, 16, 4)
	EG(2, 6, 16, 4)
	EG(2, 6, 16, 3)
	EG(4, 4, 16, 2)
#undef EG
	fprintf(stderr, "exp_engine: config %d not built\n", config);
	return -1;
}
