// exp_mega.hip -- EXPERIMENT (not product code): one persistent launch per decode step, second attempt.
//
// Round 1's persistent variants (exp_overlap.hip) tied with or lost to launch-per-kernel:
//   k_persist  48.2 us/layer  -- prefetch starts AT the edge; the edge itself is one 256-arrival counter (3-4.5 us of
//                                fan-in), a release and an acquire fence (1.7 us each) and the vector reload;
//   k_deep     75 us/layer at every depth -- the lone coordinator wave staged the vector in eight dependent
//                                load -> reduce rounds and arrived on the same single counter.
// This file keeps k_deep's split of roles (STREAMER waves whose only global-memory traffic is the weight stream, run
// DEPTH x 8 KiB ahead of their consumption across phase boundaries; HELPER waves with empty memory queues that publish,
// arrive, poll and stage) and rebuilds the edge:
//   * results leave the workgroup as ONE contiguous write-through (sc1) store burst by helper 0 (a workgroup owns a
//     contiguous block of output rows), then s_waitcnt vmcnt(0), then ONE arrival;
//   * arrivals are sharded over 8 counters (one per blockIdx % 8 = XCD; 32 arrivals each, own 128-byte lines), polled by
//     8 lanes of helper 0 with relaxed sc1 loads -- no fences anywhere;
//   * the vector is reloaded with sc1 (or sc0 sc1) 16-byte loads by ALL helpers at once, every load issued before the
//     first is waited for.
// The arithmetic per task equals k_stream's, so the final vector must match exp_chain's checksum bit for bit.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/experiments/libexp_mega.so tools/experiments/exp_mega.hip
#include "exp_overlap.hip"
#include <algorithm>

constexpr int MG_MAXP = 136;
constexpr int MG_MAXT = 16;   // tasks of one phase per streamer
constexpr int MG_SHARD = 32;  // uints between counter shards (128 bytes)

struct MPhase {
	const void* w;
	unsigned ntasks;
	unsigned K; // tasks per streamer (ceil)
};

// 16-byte load that bypasses the L1 (sc1: agent scope) or L1 and L2 (sc0 sc1: system scope); not tracked by the compiler's
// vmcnt bookkeeping -- the caller waits with mg_wait0 on the registers it is about to use
template <int SYS>
__device__ __forceinline__ f32x4 mg_load16(const float* p) {
	f32x4 v;
	if constexpr (SYS) {
		asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	} else {
		asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	}
	return v;
}
template <int SYS>
__device__ __forceinline__ void mg_store4(float* p, float v) {
	if constexpr (SYS) {
		__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	} else {
		__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

template <int NS, int NH, int DEPTH, int REAL, int SYS, int DBG>
__global__ __launch_bounds__((NS + NH) * 64) void k_mega(const MPhase* __restrict__ ph_global, int nphases, float* x0, float* x1, unsigned* cnt, unsigned* timeout,
                                                          unsigned long long* dbg) {
	__shared__ MPhase ph[MG_MAXP];
	__shared__ __attribute__((aligned(16))) float xs[VEC];
	__shared__ float red[NW];
	__shared__ float outv[NS * MG_MAXT];
	__shared__ int finished; // streamer-phases left so far (monotonic)
	__shared__ int gate;     // phases whose grid barrier helper 0 has seen pass (monotonic)
	__shared__ int staged;   // helper-phases staged so far (monotonic)
	__shared__ int gave_up;
	__shared__ unsigned long long sdbg[DBG ? MG_MAXP : 1][2];
	__shared__ unsigned long long tdbg[DBG ? 256 : 1][2]; // streamer 0: per consumed tile, clock before / after its data wait
	constexpr int BT = (NS + NH) * 64;
	for (int i = threadIdx.x; i < nphases; i += BT) {
		ph[i] = ph_global[i];
	}
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (threadIdx.x == 0) {
		finished = 0, gate = 0, staged = 0, gave_up = 0;
	}
	__syncthreads();
	const bool dbg_blk = DBG && (blockIdx.x == 0 || blockIdx.x == 137);
	unsigned long long* dbgp = dbg + (blockIdx.x == 0 ? 0 : 1) * (size_t)(MG_MAXP * 8 + 512);

	if (wave >= NS) {
		// ------------------------------------------------ helpers: empty memory queues
		const int h = wave - NS;
		for (int p = 0; p <= nphases; ++p) {
			if (p > 0) {
				if (h == 0) {
					unsigned spins = 0;
					while (__hip_atomic_load(&finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NS * p) {
						__builtin_amdgcn_s_sleep(1);
						if (++spins > (1u << 22)) {
							*timeout = 2;
							break;
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					if (dbg_blk && lane == 0) {
						dbgp[(p - 1) * 8 + 0] = wall_clock64();
					}
					// publish phase p-1: this workgroup's tasks are t = base .. base + K * NS - 1
					float* xout = ((p - 1) & 1) ? x0 : x1;
					const unsigned nt = ph[p - 1].ntasks, kn = ph[p - 1].K * NS, base = blockIdx.x * kn;
					for (unsigned i = lane; i < kn; i += 64) {
						const unsigned t = base + i;
						if (t < nt && t < VEC) {
							mg_store4<SYS>(xout + t, outv[i]);
						}
					}
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					if (lane == 0) {
						__hip_atomic_fetch_add(cnt + (blockIdx.x & 7) * MG_SHARD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					if (dbg_blk && lane == 0) {
						dbgp[(p - 1) * 8 + 1] = wall_clock64();
					}
					if (p == nphases) {
						if (lane == 0 && gave_up) {
							*timeout = (unsigned)gave_up;
						}
						if (dbg_blk && lane == 0) {
							for (int q = 0; q < nphases; ++q) {
								dbgp[q * 8 + 5] = sdbg[q][0];
								dbgp[q * 8 + 6] = sdbg[q][1];
							}
							for (int q = 0; q < 256; ++q) {
								dbgp[MG_MAXP * 8 + q * 2] = tdbg[q][0];
								dbgp[MG_MAXP * 8 + q * 2 + 1] = tdbg[q][1];
							}
						}
						break;
					}
					const unsigned target = (gridDim.x / 8) * (unsigned)p;
					spins = 0;
					for (;;) {
						unsigned v = target;
						if (lane < 8) {
							v = __hip_atomic_load(cnt + lane * MG_SHARD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
					if (dbg_blk && lane == 0) {
						dbgp[(p - 1) * 8 + 2] = wall_clock64();
					}
					if (lane == 0) {
						__hip_atomic_store(&gate, p, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				} else {
					if (p == nphases) {
						break;
					}
					unsigned spins = 0;
					while (__hip_atomic_load(&gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < p) {
						__builtin_amdgcn_s_sleep(1);
						if (++spins > (1u << 22)) {
							break;
						}
					}
				}
			}
			// stage the vector of phase p.  Sum of squares in k_stream's order: emulated wave wv of a 512-thread block
			// (threads wv * 64 + lane hold float4 #tid and #tid + 512), one butterfly per emulated wave, totals added in
			// wave order by the consumers.
			const float* xin = (p & 1) ? x1 : x0;
			constexpr int R = (NW + NH - 1) / NH;
			f32x4 a[R], b[R];
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const int wv = h + r * NH;
				const int tid = (wv < NW ? wv : 0) * 64 + lane;
				a[r] = mg_load16<SYS>(xin + 4 * tid);
				b[r] = mg_load16<SYS>(xin + 4 * (tid + BLOCK));
			}
#pragma unroll
			for (int r = 0; r < R; ++r) {
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[r]), "+v"(b[r])::"memory");
			}
			if (dbg_blk && h == 0 && lane == 0 && p > 0) {
				dbgp[(p - 1) * 8 + 3] = wall_clock64();
			}
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const int wv = h + r * NH;
				if (wv < NW) {
					const int tid = wv * 64 + lane;
					((f32x4*)xs)[tid] = a[r];
					((f32x4*)xs)[tid + BLOCK] = b[r];
					float ss = a[r].x * a[r].x + a[r].y * a[r].y + a[r].z * a[r].z + a[r].w * a[r].w + b[r].x * b[r].x + b[r].y * b[r].y + b[r].z * b[r].z +
					           b[r].w * b[r].w;
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
			if (dbg_blk && h == 0 && lane == 0 && p > 0) {
				dbgp[(p - 1) * 8 + 4] = wall_clock64();
			}
		}
		return;
	}

	// ---------------------------------------------------- streamers: weight loads only
	// Cursor over this streamer's (phase, task) sequence; the phase descriptor is re-read from LDS on a phase change only.
	struct Cur {
		const unsigned char* w;
		unsigned ntasks, K, k, t;
		int p;
	};
	const int s = wave;
	auto enter = [&](Cur& c) { // position c at the first task of phase c.p that this streamer owns, skipping empty phases
		for (; c.p < nphases; ++c.p) {
			c.w = (const unsigned char*)ph[c.p].w;
			c.ntasks = ph[c.p].ntasks;
			c.K = ph[c.p].K;
			c.k = 0;
			c.t = blockIdx.x * c.K * NS + s;
			if (c.t < c.ntasks) {
				return;
			}
		}
	};
	u32x4 tile[DEPTH][8];
	Cur ic, cc;
	ic.p = 0;
	enter(ic);
	cc = ic;
	const unsigned char* wlast = (const unsigned char*)ph[nphases - 1].w;
	auto issue = [&](auto KK) {
		constexpr int k = decltype(KK)::value;
		const bool live = ic.p < nphases;
		const unsigned char* src = live ? ic.w + (size_t)ic.t * 8192 : wlast; // past the end: re-read something valid, drop it
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[k][u] = __builtin_nontemporal_load((gptr16)src + u * 64 + lane);
		}
		if (live) {
			++ic.k;
			ic.t += NS;
			if (__builtin_expect(ic.k >= ic.K || ic.t >= ic.ntasks, 0)) {
				++ic.p;
				enter(ic);
			}
		}
	};
	auto first = [&](auto KK) {
		issue(KK);
		return false;
	};
	static_steps<0, DEPTH>(first);
	int ntile = 0, left = 0; // phases handed over so far
	float scale = 0.f;
	auto hand_over = [&](int upto) { // this streamer is done with every phase < upto
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0) {
			if (DBG && s == 0) {
				const unsigned long long now = wall_clock64();
				for (int q = left; q < upto; ++q) {
					sdbg[q][1] = now;
				}
			}
			__hip_atomic_fetch_add(&finished, upto - left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		left = upto;
	};
	auto wait_staged = [&](int p) {
		unsigned spins = 0;
		while (__hip_atomic_load(&staged, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < NH * (p + 1)) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > (1u << 22)) {
				gave_up = 4;
				break;
			}
		}
		float tot = 0.f;
#pragma unroll
		for (int i = 0; i < NW; ++i) {
			tot += red[i];
		}
		scale = 1.0f / sqrtf(tot / VEC + 1e-5f);
		if (DBG && s == 0 && lane == 0) {
			sdbg[p][0] = wall_clock64();
		}
	};
	if (cc.p < nphases) {
		hand_over(cc.p); // phases before the first one with work
		wait_staged(cc.p);
	}
	auto step = [&](auto KK) -> bool { // consume slot KK, refill it; true when this streamer has no task left
		constexpr int k = decltype(KK)::value;
		if (cc.p >= nphases) {
			return true;
		}
		const unsigned t = cc.t;
		const float xsv = xs[(t * 7) % VEC];
		unsigned long long tw0 = 0;
		if (DBG && s == 0) {
			tw0 = wall_clock64();
		}
		const float tv = tile_value<REAL>(tile[k], xs, lane);
		if (DBG && s == 0) {
			asm volatile("" ::"v"(tv));
			const unsigned long long tw1 = wall_clock64();
			if (lane == 0 && ntile < 256) {
				tdbg[ntile][0] = tw0;
				tdbg[ntile][1] = tw1;
			}
			++ntile;
		}
		issue(KK); // the slot is free again: DEPTH tasks ahead, whatever phase that is
		const float v = wave_sum(tv) * scale * xsv;
		if (lane == 0) {
			outv[cc.k * NS + s] = v + (float)(t % 13);
		}
		++cc.k;
		cc.t += NS;
		if (__builtin_expect(cc.k >= cc.K || cc.t >= cc.ntasks, 0)) {
			++cc.p;
			enter(cc);
			hand_over(cc.p < nphases ? cc.p : nphases);
			if (cc.p < nphases) {
				wait_staged(cc.p);
			}
		}
		return false;
	};
	while (!static_steps<0, DEPTH>(step)) {
	}
	if (left < nphases) {
		hand_over(nphases);
	}
}

template <int NS, int NH, int DEPTH, int REAL, int SYS, int DBG>
static double mega_run(int n_layers, int iters, double* checksum, int print_dbg) {
	static const size_t sizes[4] = {25165824, 16777216, 117440512, 58720256};
	const int NK = 4, total = n_layers * NK, grid = 256;
	if (total > MG_MAXP) {
		fprintf(stderr, "exp_mega: at most %d phases\n", MG_MAXP);
		return -1;
	}
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	std::vector<void*> w(total);
	std::vector<MPhase> hp(total);
	for (int i = 0; i < total; ++i) {
		CK(hipMalloc(&w[i], sizes[i % NK] + 65536));
		CK(hipMemset(w[i], 0x11 + i / NK + i % NK, sizes[i % NK] + 65536));
		hp[i].w = w[i];
		hp[i].ntasks = (unsigned)(sizes[i % NK] / 8192);
		hp[i].K = (hp[i].ntasks + grid * NS - 1) / (grid * NS);
		if (hp[i].K > MG_MAXT) {
			fprintf(stderr, "exp_mega: %u tasks per streamer exceed %d\n", hp[i].K, MG_MAXT);
			return -1;
		}
	}
	MPhase* dp;
	CK(hipMalloc(&dp, sizeof(MPhase) * total));
	CK(hipMemcpy(dp, hp.data(), sizeof(MPhase) * total, hipMemcpyHostToDevice));
	float* xbuf[2];
	CK(hipMalloc(&xbuf[0], VEC * 4 + 65536));
	CK(hipMalloc(&xbuf[1], VEC * 4 + 65536));
	std::vector<float> x0(VEC);
	for (int i = 0; i < VEC; ++i) {
		x0[i] = 0.001f * (i % 97) + 0.5f;
	}
	unsigned *cnt, *timeout;
	unsigned long long* dbg;
	CK(hipMalloc(&cnt, 4 * MG_SHARD * 8));
	CK(hipMalloc(&timeout, 4));
	CK(hipMemset(timeout, 0, 4));
	CK(hipMalloc(&dbg, 2 * (MG_MAXP * 8 + 512) * 8));
	CK(hipMemset(dbg, 0, 2 * (MG_MAXP * 8 + 512) * 8));
	auto run = [&]() {
		CK(hipMemcpyAsync(xbuf[0], x0.data(), VEC * 4, hipMemcpyHostToDevice, s));
		CK(hipMemsetAsync(cnt, 0, 4 * MG_SHARD * 8, s));
		hipLaunchKernelGGL((k_mega<NS, NH, DEPTH, REAL, SYS, DBG>), dim3(grid), dim3((NS + NH) * 64), 0, s, (const MPhase*)dp, total, xbuf[0], xbuf[1], cnt, timeout,
		                   dbg);
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
		printf("  !! a bounded spin timed out (mega NS=%d NH=%d DEPTH=%d, code %u)\n", NS, NH, DEPTH, to);
	}
	std::vector<unsigned long long> hd_keep;
	if (DBG && print_dbg) {
		std::vector<unsigned long long> hd(2 * (MG_MAXP * 8 + 512));
		CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
		// per phase kind, averaged over layers >= 1: microseconds from "all local streamers left the phase" (t0)
		const char* names[4] = {"qkv", "wo", "ffn_up", "ffn_down"};
		for (int blk = 0; blk < 2; ++blk) {
			printf("  block %d: edge after phase kind | publish+arrive | barrier wait | vector load | stage | streamer0 sees staged (all from t0) | streamer0 busy in next phase | helper idle before t0\n",
			       blk ? 137 : 0);
			for (int kind = 0; kind < 4; ++kind) {
				double acc[7] = {0, 0, 0, 0, 0, 0, 0};
				int n = 0;
				for (int p = NK + kind; p + 1 < total; p += NK) {
					const unsigned long long* d = &hd[(size_t)blk * (MG_MAXP * 8 + 512) + (size_t)p * 8];
					const unsigned long long* dn = &hd[(size_t)blk * (MG_MAXP * 8 + 512) + (size_t)(p + 1) * 8];
					const unsigned long long* dp_ = &hd[(size_t)blk * (MG_MAXP * 8 + 512) + (size_t)(p - 1) * 8];
					acc[0] += (double)(d[1] - d[0]);
					acc[1] += (double)(d[2] - d[0]);
					acc[2] += (double)(d[3] - d[0]);
					acc[3] += (double)(d[4] - d[0]);
					acc[4] += (double)((long long)(dn[5] - d[0]));
					acc[5] += (double)((long long)(dn[6] - dn[5]));
					acc[6] += (double)((long long)(d[0] - dp_[4]));
					++n;
				}
				printf("    %-9s %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f | %6.2f\n", names[kind], acc[0] / n / 100, acc[1] / n / 100, acc[2] / n / 100, acc[3] / n / 100,
				       acc[4] / n / 100, acc[5] / n / 100, acc[6] / n / 100);
			}
		}
		hd_keep = hd;
	}
	if (DBG && print_dbg && NS == 4) {
		const unsigned long long* td = &hd_keep[(size_t)(MG_MAXP * 8)];
		const unsigned long long* d0 = &hd_keep[0];
		printf("  block 0 streamer 0, layer 2: tile# kind | wait start (us since layer's first staged) | wait+consume us\n");
		const unsigned long long origin = d0[(2 * NK) * 8 + 5] ? d0[(2 * NK) * 8 + 5] : td[52 * 2];
		for (int i = 52; i < 78 + 5; ++i) {
			printf("    %3d  %8.2f  %6.2f\n", i - 52, (double)((long long)(td[i * 2] - origin)) / 100, (double)(td[i * 2 + 1] - td[i * 2]) / 100);
		}
		for (int p = 2 * NK; p < 3 * NK + 1; ++p) {
			printf("    phase %d: staged seen %8.2f   left %8.2f  (helper: t0 %8.2f published %8.2f passed %8.2f staged %8.2f)\n", p, (double)((long long)(d0[p * 8 + 5] - origin)) / 100,
			       (double)((long long)(d0[p * 8 + 6] - origin)) / 100, (double)((long long)(d0[p * 8 + 0] - origin)) / 100, (double)((long long)(d0[p * 8 + 1] - origin)) / 100,
			       (double)((long long)(d0[p * 8 + 2] - origin)) / 100, (double)((long long)(d0[p * 8 + 4] - origin)) / 100);
		}
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
	CK(hipFree(dbg));
	CK(hipStreamDestroy(s));
	return (double)ms * 1e3 / ((double)iters * n_layers);
}

// config = NS * 1000 + NH * 100 + DEPTH * 10 + SYS;  flags: bit 0 REAL, bit 1 DBG
extern "C" double exp_mega(int config, int flags, int n_layers, int iters, double* checksum) {
#define MG(ns, nh, d, sys)                                                                 \
	if (config == ns * 1000 + nh * 100 + d * 10 + sys) {                                   \
		switch (flags & 3) {                                                               \
		case 0:                                                                            \
			return mega_run<ns, nh, d, 0, sys, 0>(n_layers, iters, checksum, 0);           \
		case 1:                                                                            \
			return mega_run<ns, nh, d, 1, sys, 0>(n_layers, iters, checksum, 0);           \
		case 2:                                                                            \
			return mega_run<ns, nh, d, 0, sys, 1>(n_layers, iters, checksum, 1);           \
		default:                                                                           \
			return mega_run<ns, nh, d, 1, sys, 1>(n_layers, iters, checksum, 1);           \
		}                                                                                  \
	}
	MG(4, 1, 4, 0)
	MG(4, 1, 6, 0)
	MG(4, 4, 4, 0)
	MG(4, 4, 6, 0)
	MG(4, 4, 4, 1)
	MG(4, 4, 6, 1)
	MG(4, 2, 6, 0)
	MG(7, 1, 4, 0)
	MG(6, 2, 4, 0)
	MG(4, 4, 2, 0)
#undef MG
	fprintf(stderr, "exp_mega: config %d not built\n", config);
	return -1;
}

// ---- burst probe: every wave issues DEPTH x 8 KiB at once and then only waits.  When does tile k land?  (Is the data of a
// deep one-shot prefetch delivered at the full HBM rate while the waves sit idle, as an edge inside a launch needs?)
template <int DEPTH>
__global__ __launch_bounds__(512) void k_burst(const void* w, unsigned long long* out, unsigned* sink) {
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int nw = blockDim.x >> 6;
	u32x4 tile[DEPTH][8];
	const size_t wid = (size_t)blockIdx.x * nw + wave;
	const unsigned long long t0 = wall_clock64();
#pragma unroll
	for (int k = 0; k < DEPTH; ++k) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[k][u] = __builtin_nontemporal_load((gptr16)w + (wid * DEPTH + k) * 512 + u * 64 + lane);
		}
	}
	const unsigned long long t1 = wall_clock64();
	unsigned long long tl[DEPTH];
	unsigned acc = 0;
#pragma unroll
	for (int k = 0; k < DEPTH; ++k) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			acc += tile[k][u][0] ^ tile[k][u][3];
		}
		asm volatile("" ::"v"(acc));
		tl[k] = wall_clock64();
	}
	if (acc == 0x12345678u) {
		*sink = acc;
	}
	if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 137 || blockIdx.x == 255)) {
		unsigned long long* o = out + ((blockIdx.x == 0 ? 0 : (blockIdx.x == 137 ? 1 : 2)) * 8 + wave) * 16;
		o[0] = t0;
		o[1] = t1;
#pragma unroll
		for (int k = 0; k < DEPTH; ++k) {
			o[2 + k] = tl[k];
		}
	}
}

extern "C" void exp_burst(int depth, int ns) {
	const size_t bytes = (size_t)256 * ns * depth * 8192;
	void* w;
	CK(hipMalloc(&w, bytes * 4 + 65536));
	CK(hipMemset(w, 0x5a, bytes * 4 + 65536));
	unsigned long long* out;
	unsigned* sink;
	CK(hipMalloc(&out, 3 * 8 * 16 * 8));
	CK(hipMalloc(&sink, 4));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	float ms = 0;
	for (int rep = 0; rep < 4; ++rep) { // a different quarter of the buffer each time: nothing comes from a cache
		const void* wr = (const char*)w + rep * bytes;
		CK(hipMemset(out, 0, 3 * 8 * 16 * 8));
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0, 0));
		switch (depth) {
		case 2:
			hipLaunchKernelGGL(k_burst<2>, dim3(256), dim3(ns * 64), 0, 0, wr, out, sink);
			break;
		case 4:
			hipLaunchKernelGGL(k_burst<4>, dim3(256), dim3(ns * 64), 0, 0, wr, out, sink);
			break;
		case 6:
			hipLaunchKernelGGL(k_burst<6>, dim3(256), dim3(ns * 64), 0, 0, wr, out, sink);
			break;
		default:
			hipLaunchKernelGGL(k_burst<12>, dim3(256), dim3(ns * 64), 0, 0, wr, out, sink);
			depth = 12;
		}
		CK(hipEventRecord(e1, 0));
		CK(hipDeviceSynchronize());
		CK(hipEventElapsedTime(&ms, e0, e1));
	}
	std::vector<unsigned long long> h(3 * 8 * 16);
	CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
	printf("burst depth %d x 8 KiB, %d waves/CU (%.1f MB chip-wide; %.2f us at 6.3 TB/s; kernel %.2f us):\n", depth, ns, bytes / 1e6, bytes / 6.3e6, ms * 1e3);
	for (int b = 0; b < 3; ++b) {
		for (int wv = 0; wv < ns; wv += (ns > 4 ? 3 : 1)) {
			const unsigned long long* o = &h[(b * 8 + wv) * 16];
			printf("  block %3d wave %d: issue done %5.2f | landed", b == 0 ? 0 : (b == 1 ? 137 : 255), wv, (double)(o[1] - o[0]) / 100);
			for (int k = 0; k < depth; ++k) {
				printf(" %5.2f", (double)(o[2 + k] - o[0]) / 100);
			}
			printf("\n");
		}
	}
	fflush(stdout);
	CK(hipFree(w));
	CK(hipFree(out));
	CK(hipFree(sink));
}

// ---- start-up probe: what delays a streaming kernel's FIRST tile?  Every wave of a 256-thread workgroup (GRIDxWPC) does what the
// product's row engine does at launch, in variants:  mode 0: tile loads only;  mode 1: first the loads of a shared 16 KiB vector
// (+ 16 KiB of "norm weights"), then the tile, then the vector is reduced block-wide and written to LDS (two barriers) before the
// tile is touched;  mode 2: as 1, with a second tile issued before the prologue (the product's order until round 2).
template <int MODE>
__global__ __launch_bounds__(256) void k_startup(const void* w, const float* x, unsigned long long* out, unsigned* sink) {
	__shared__ float xs[8192];
	__shared__ float red[4];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const size_t wid = (size_t)blockIdx.x * 4 + wave;
	const unsigned long long t0 = wall_clock64();
	float4 xv[4], gv[4];
	if (MODE >= 1) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			xv[i] = ((const float4*)x)[threadIdx.x + i * 256];
			gv[i] = ((const float4*)x)[4096 + threadIdx.x + i * 256];
		}
	}
	u32x4 tile[2][8];
#pragma unroll
	for (int u = 0; u < 8; ++u) {
		tile[0][u] = __builtin_nontemporal_load((gptr16)w + wid * 1024 + u * 64 + lane);
	}
	if (MODE == 2) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[1][u] = __builtin_nontemporal_load((gptr16)w + wid * 1024 + 512 + u * 64 + lane);
		}
	}
	const unsigned long long t1 = wall_clock64(); // loads issued
	unsigned long long t2 = t1, t3 = t1;
	float scale = 1.f;
	if (MODE >= 1) {
		float ss = 0.f;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
		}
		asm volatile("" ::"v"(ss));
		t2 = wall_clock64(); // vector landed
		ss = wave_sum(ss);
		if (lane == 0) {
			red[wave] = ss;
		}
		__syncthreads();
		scale = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / 4096.f + 1e-5f);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			((float4*)xs)[threadIdx.x + i * 256] = make_float4(xv[i].x * scale * gv[i].x, xv[i].y * scale * gv[i].y, xv[i].z * scale * gv[i].z, xv[i].w * scale * gv[i].w);
		}
		__syncthreads();
		t3 = wall_clock64(); // image built
	}
	unsigned acc = 0;
#pragma unroll
	for (int u = 0; u < 8; ++u) {
		acc += tile[0][u][0] ^ tile[0][u][3];
	}
	asm volatile("" ::"v"(acc));
	const unsigned long long t4 = wall_clock64(); // first tile landed
	if (MODE == 2) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			acc += tile[1][u][1];
		}
	}
	if (acc == 0x12345678u || scale == 12345.f) {
		*sink = acc + (unsigned)xs[lane];
	}
	if (lane == 0 && wid < 4096) {
		unsigned long long* o = out + wid * 8;
		o[0] = t0, o[1] = t1, o[2] = t2, o[3] = t3, o[4] = t4;
	}
}

extern "C" void exp_startup(int mode, int grid) {
	const size_t bytes = (size_t)grid * 4 * 16384;
	void* w;
	float* x;
	CK(hipMalloc(&w, bytes * 4 + 65536));
	CK(hipMemset(w, 0x5a, bytes * 4 + 65536));
	CK(hipMalloc(&x, 65536 + 65536));
	CK(hipMemset(x, 0, 65536 + 65536));
	unsigned long long* out;
	unsigned* sink;
	CK(hipMalloc(&out, 4096 * 8 * 8));
	CK(hipMalloc(&sink, 4));
	std::vector<unsigned long long> h(4096 * 8);
	double acc[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
	const int reps = 4;
	for (int rep = 0; rep < reps; ++rep) {
		const void* wr = (const char*)w + rep * bytes;
		CK(hipMemset(out, 0, 4096 * 8 * 8));
		CK(hipMemset(x, 0x3c, 65536)); // the vector is freshly written, like a producer kernel's output
		CK(hipDeviceSynchronize());
		if (mode == 0) {
			hipLaunchKernelGGL(k_startup<0>, dim3(grid), dim3(256), 0, 0, wr, x, out, sink);
		} else if (mode == 1) {
			hipLaunchKernelGGL(k_startup<1>, dim3(grid), dim3(256), 0, 0, wr, x, out, sink);
		} else {
			hipLaunchKernelGGL(k_startup<2>, dim3(grid), dim3(256), 0, 0, wr, x, out, sink);
		}
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
		unsigned long long t0 = ~0ull;
		const int nw = grid * 4 < 4096 ? grid * 4 : 4096;
		for (int i = 0; i < nw; ++i) {
			t0 = h[i * 8] < t0 ? h[i * 8] : t0;
		}
		std::vector<double> v[4];
		for (int i = 0; i < nw; ++i) {
			for (int k = 0; k < 4; ++k) {
				v[k].push_back((double)(h[i * 8 + 1 + k] - t0) / 100);
			}
		}
		for (int k = 0; k < 4; ++k) {
			std::sort(v[k].begin(), v[k].end());
			acc[k] += v[k][v[k].size() / 2];
			mx[k] += v[k].back();
		}
	}
	printf("startup mode %d, %d workgroups x 4 waves (%.1f MB per tile round): loads issued p50 %.2f max %.2f | vector landed %.2f / %.2f | image built %.2f / %.2f | first tile landed %.2f / %.2f us\n",
	       mode, grid, (double)grid * 4 * 8192 / 1e6, acc[0] / reps, mx[0] / reps, acc[1] / reps, mx[1] / reps, acc[2] / reps, mx[2] / reps, acc[3] / reps, mx[3] / reps);
	fflush(stdout);
	CK(hipFree(w));
	CK(hipFree(x));
	CK(hipFree(out));
	CK(hipFree(sink));
}
