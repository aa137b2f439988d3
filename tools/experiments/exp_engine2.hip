// exp_engine2.hip -- EXPERIMENT (not product code): the LDS-DMA loader/consumer engine, second cut -- round 3's time-boxed gate.
//
// Round 2's first cut (exp_engine.hip) lost to launch-per-kernel by 8-13 %: its loader streamed at 5.0 TB/s instead of 6.4, and
// every edge waited on arrival counters polled by one wave per workgroup.  This cut follows the CDNA guide's recipe row by row
// (MI355X_MICROARCH.md, price list: ldsdma-fill, nt-weights, allgather, gather-pass, transport-variants, engine-vs-launches):
//   * one LOADER wave + 3 CONSUMER waves per workgroup, one workgroup per CU;
//   * 16 KiB ring slots (16 x 1-KiB global_load_lds_dwordx4, non-temporal), ring of R slots, at most D fills in flight; one M0
//     write per four DMA pieces (instruction offsets 0 / 1024 / 2048 / 3072 move both addresses);
//   * hand-offs are 8-byte {value, tag} GRANULES written by ONE sc1 store each, fire and forget -- no counters, no arrivals, no
//     fences: a consumer whose phase is over sweeps the granule vector (sc1 dwordx2 loads, 32 per lane per pass), compares tags,
//     re-reads what is not there yet, writes the payload to LDS; the first consumer of a workgroup to leave a phase does the
//     sweep for the workgroup while the others finish their slots;
//   * while its CU sweeps, the loader is thinned to one fill in flight (its DMA otherwise queues ahead of the sweep's loads).
// The chain is exp_overlap.hip's: 4 streaming phases per layer (25.2 / 16.8 / 117.4 / 58.7 MB), every 8 KiB task t < 4096
// produces one float of the next phase's input vector, every workgroup needs the whole vector and its sum of squares.
// k_ref_phase computes the same chain with one plain launch per phase over the same granule buffers: the engine's final vector
// must equal it bit for bit.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/experiments/libexp_engine2.so tools/experiments/exp_engine2.hip
#include "exp_overlap.hip"

#include <string.h>

constexpr int E2_SLOT = 16384;   // bytes per ring slot = two 8 KiB tasks
constexpr int E2_MAXP = 136;     // phases per launch
constexpr int E2_NC = 3;         // consumer waves
constexpr unsigned E2_SPIN = 1u << 21;

struct E2Phase {
	const void* w;
	unsigned nslots; // 16 KiB slots of this phase (ntasks / 2)
	unsigned pad;
};
// Workgroup b takes slots b, b + grid, b + 2 grid, ... of a phase: at any moment the chip reads ONE moving window of consecutive
// slots (every memory channel busy), as the launch-per-kernel chain does.  (First cut: a contiguous run of slots per workgroup --
// 256 streams a fixed stride apart; it never got past 4.8 TB/s with the edges switched off.)
__device__ __forceinline__ int e2_count(unsigned nslots, unsigned b, unsigned grid) {
	return nslots > b ? (int)((nslots - b + grid - 1) / grid) : 0;
}
// task t of a phase with ntasks tasks writes entry ntasks - 1 - t of the next vector if that is < VEC: the LAST tasks of a phase
// produce its outputs, so an edge completes when the phase does (as every row of a real matvec phase produces an output)
__device__ __forceinline__ unsigned e2_out_index(unsigned t, unsigned ntasks) {
	return ntasks - 1u - t;
}

typedef __attribute__((address_space(3))) void* e2_lds_t;
__device__ __forceinline__ unsigned e2_lds_addr(const void* p) {
	return (unsigned)(size_t)(e2_lds_t)p;
}
// LDS accesses the compiler must not see in the loader (it would order each behind the pending LDS-DMA with vmcnt(0))
__device__ __forceinline__ int e2_lds_read(unsigned addr) {
	int v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return v;
}
__device__ __forceinline__ unsigned long long e2_lds_read64(unsigned addr) {
	unsigned long long v;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return v;
}
__device__ __forceinline__ void e2_lds_write(unsigned addr, int v) {
	asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void e2_vmcnt() {
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one granule: {payload, tag}, one 8-byte write-through store; never waited for
__device__ __forceinline__ void e2_publish(uint2* g, float v, unsigned tag) { // (agent-scope relaxed: global_store_dwordx2 sc1)
	__hip_atomic_store((unsigned long long*)g, (unsigned long long)__builtin_bit_cast(unsigned, v) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int E2_NOEDGE = 1; // flags bit 0: tags are not checked (diagnostic: wrong results; the in-phase rate only)
constexpr int E2_THIN = 2;   // flags bit 1: thin the loader to one fill in flight while the workgroup sweeps
constexpr int E2_TREE = 4;   // flags bit 2: two-level gather -- workgroup g < 8 sweeps the producers' granules for group g (the workgroups
                             // b with b % 8 == g: one XCD, as blocks are observed to be placed) and re-publishes the vector into the
                             // group's staging buffer, which the other 31 sweep: 8 readers of the hot 32 KiB instead of 256, and 31
                             // readers each of a copy their own L2 holds

// the loader wave.  Two cursors: `issued` fills handed to the DMA, `published` fills whose landing the consumers have been told
// about.  A fill is issued whenever its ring slot is free and fewer than D are in flight; otherwise the loader waits for its
// OLDEST fill in flight (counted vmcnt) and publishes it.  (First form: a landing was only published on the way to the next issue,
// so a loader held up by a full ring also held back the last D - 1 landed fills from the consumers it was waiting for.)
template <int R, int D>
__device__ __forceinline__ void e2_loader(unsigned a_ph, unsigned a_cum, unsigned a_ring, unsigned a_landed, unsigned a_done, unsigned a_sweeping, unsigned a_gaveup, unsigned a_stalls, int G, int lane,
                                          bool thin_on) {
	static_assert(D >= 1 && D <= 4, "vmcnt holds 63: at most 3 older fills of 16 pieces behind the newest");
	int p = -1, p_end = 0, p_begin = 0;
	const unsigned char* wbase = nullptr;
	unsigned stalls = 0; // polls of a full ring (diagnostic)
	int issued = 0, published = 0;
	unsigned spins = 0;
	while (published < G) {
		bool can_issue = issued < G && issued - published < D;
		if (can_issue && thin_on && issued - published >= 1 && e2_lds_read(a_sweeping) != 0) {
			can_issue = false; // one fill in flight only while this CU sweeps granules
		}
		if (can_issue && issued >= R) { // the slot's previous tenant (fill issued - R, consumer (issued - R) % NC, its ((issued - R) / NC)-th) must be consumed
			const int c = (issued - R) % E2_NC, j = (issued - R) / E2_NC;
			if (e2_lds_read(a_done + 4 * c) <= j) {
				can_issue = false;
				++stalls;
				if (issued == published) { // nothing in flight to wait for: poll
					__builtin_amdgcn_s_sleep(1);
					if (++spins > E2_SPIN) {
						e2_lds_write(a_gaveup, 5);
						return;
					}
					continue;
				}
			}
		}
		if (can_issue) {
			const int g = issued;
			while (g >= p_end) { // the next phase with slots of this workgroup
				++p;
				p_begin = e2_lds_read(a_cum + 4 * p);
				p_end = e2_lds_read(a_cum + 4 * (p + 1));
				wbase = (const unsigned char*)e2_lds_read64(a_ph + 16 * p);
			}
			const unsigned s = blockIdx.x + gridDim.x * (unsigned)(g - p_begin);
			const unsigned char* src = wbase + (size_t)s * E2_SLOT + lane * 16;
			const unsigned slot = a_ring + (unsigned)(g % R) * E2_SLOT;
#pragma unroll
			for (int q = 0; q < 4; ++q) { // four pieces per M0 value: the instruction offset moves the global AND the LDS address
				const __attribute__((address_space(1))) void* gp = (const __attribute__((address_space(1))) void*)(src + q * 4096);
				const e2_lds_t lp = (e2_lds_t)(size_t)(slot + q * 4096);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 2 /* nt */);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 2);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 2);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 2);
			}
			++issued;
			spins = 0;
			continue;
		}
		// wait for the oldest fill in flight: issued - published of them are, each 16 DMA pieces
		const int inflight = issued - published;
		if (inflight >= 4) {
			e2_vmcnt<48>();
		} else if (inflight == 3) {
			e2_vmcnt<32>();
		} else if (inflight == 2) {
			e2_vmcnt<16>();
		} else {
			e2_vmcnt<0>();
		}
		++published;
		if (lane == 0) {
			e2_lds_write(a_landed, published);
		}
	}
	if (lane == 0) {
		e2_lds_write(a_stalls, (int)stalls);
	}
}

// which phase produced entry i of phase p's input, as a tag offset: phase p - 1 for the entries it writes, else phase p - 3
// (the chain's 8 KiB tasks t >= its count leave the slot to the phase two before, same buffer); tags are tag_base + phase + 4
__device__ __forceinline__ unsigned e2_expected(int i, int p, unsigned nout_prev, unsigned tag_base) {
	return tag_base + 4u + (unsigned)(i < (int)nout_prev ? p - 1 : p - 3);
}

// what the consumer-side helpers share (all LDS pointers; plain values otherwise)
struct E2C {
	float* xs;
	const E2Phase* ph;
	int* leftp;
	int* xready;
	int* sweeping;
	int* gave_up;
	float* scale_s;
	const uint2* x0;
	const uint2* x1;
	unsigned tag_base;
	int nphases, lane, noedge;
	unsigned long long* stamps;
	uint2* stage; // [8 groups][2][VEC] granules, or nullptr: flat gather
};

// -> false when the bounded spin gave up
__device__ __forceinline__ bool e2_wait_ge(int* what, int target, int* gave_up, int code, unsigned* polls = nullptr) {
	unsigned spins = 0;
	while (__hip_atomic_load(what, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
		__builtin_amdgcn_s_sleep(1);
		if (++spins > E2_SPIN) {
			*gave_up = code;
			return false;
		}
	}
	if (polls) {
		*polls += spins;
	}
	return true;
}

// sweep the granule vector that feeds phase p into xs (this wave alone), then the sum of squares in k_stream's order
__device__ __forceinline__ bool e2_sweep(const E2C& C, int p) {
	const uint2* xin = (p & 1) ? C.x1 : C.x0;
	unsigned nout_prev = p == 0 ? (unsigned)VEC : (2u * C.ph[p - 1].nslots < (unsigned)VEC ? 2u * C.ph[p - 1].nslots : (unsigned)VEC);
	const int lane = C.lane;
	const unsigned long long t_begin = C.stamps ? wall_clock64() : 0;
	// two-level gather: followers read their group's staging copy, every entry of which carries the staging tag of phase p
	const bool tree = C.stage != nullptr, leader = tree && blockIdx.x < 8;
	uint2* const stage = tree ? C.stage + ((size_t)(blockIdx.x % 8) * 2 + (p & 1)) * VEC : nullptr;
	unsigned tag_base = C.tag_base;
	if (tree && !leader) {
		xin = stage;
		nout_prev = (unsigned)VEC;
		tag_base = C.tag_base + 0x800u + 1u; // e2_expected(i < VEC, p) = tag_base + 4 + p - 1  ==  C.tag_base + 0x800 + 4 + p
	}
	__hip_atomic_store(C.sweeping, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	unsigned gr[64];              // payloads of the granules that carried the expected tag
	unsigned long long need = ~0ull; // bit j: granule j of this lane still missing
	unsigned rounds = 0;
	for (;;) {
		unsigned long long fresh[64]; // {payload, tag} (agent-scope relaxed 8-byte loads: global_load_dwordx2 sc1), missing ones only
#pragma unroll
		for (int j = 0; j < 64; ++j) {
			fresh[j] = 0;
			if (need >> j & 1ull) {
				fresh[j] = __hip_atomic_load((const unsigned long long*)(xin + j * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
#pragma unroll
		for (int j = 0; j < 64; ++j) {
			const int i = j * 64 + lane;
			const bool ok = C.noedge || (unsigned)(fresh[j] >> 32) == e2_expected(i, p, nout_prev, tag_base);
			const bool take = (need >> j & 1ull) && ok;
			gr[j] = take ? (unsigned)fresh[j] : gr[j];
			need = take ? need & ~(1ull << j) : need;
		}
		++rounds;
		if (__builtin_amdgcn_ballot_w64(need != 0) == 0) {
			break;
		}
		if (rounds > (1u << 14)) {
			*C.gave_up = 9;
			return false;
		}
		__builtin_amdgcn_s_sleep(16); // ~0.4 us between polls: pollers beside a weight stream cost it bandwidth
	}
	if (leader) { // the group's copy: the same payloads under the staging tag (write-through, never waited for)
#pragma unroll
		for (int j = 0; j < 64; ++j) {
			__hip_atomic_store((unsigned long long*)(stage + j * 64 + lane), (unsigned long long)gr[j] | ((unsigned long long)(C.tag_base + 0x800u + 4u + (unsigned)p) << 32),
			                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	const unsigned long long t_seen = C.stamps ? wall_clock64() : 0;
	if (p > 0 && !e2_wait_ge(&C.leftp[p - 1], E2_NC, C.gave_up, 10)) { // (phase p - 1 is over for every consumer: xs is free)
		return false;
	}
#pragma unroll
	for (int j = 0; j < 64; ++j) {
		C.xs[j * 64 + lane] = __builtin_bit_cast(float, gr[j]);
	}
	__hip_atomic_store(C.sweeping, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // xs is written (a wave's LDS operations complete in order)
	// sum of squares as k_stream's eight waves form it (two float4 per thread of a 512-thread block), folded in wave order
	float tot = 0.f;
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		const f32x4 a = ((const f32x4*)C.xs)[w * 64 + lane], b = ((const f32x4*)C.xs)[w * 64 + lane + BLOCK];
		const float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
		tot += wave_sum(ss);
	}
	if (lane == 0) {
		*C.scale_s = 1.0f / sqrtf(tot / VEC + 1e-5f);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	if (lane == 0) {
		__hip_atomic_store(C.xready, p + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (C.stamps) { // [phase][workgroup][4]: sweep begun, every tag seen, vector staged, polling rounds
			unsigned long long* o = C.stamps + ((size_t)p * gridDim.x + blockIdx.x) * 4;
			o[0] = t_begin, o[1] = t_seen, o[2] = wall_clock64(), o[3] = rounds;
		}
	}
	return true;
}

// this consumer is through with phase p: the first one out sweeps phase p + 1's vector for the workgroup
__device__ __forceinline__ bool e2_leave_phase(const E2C& C, int p) {
	int before = 0;
	if (C.lane == 0) {
		before = __hip_atomic_fetch_add(&C.leftp[p], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	before = __builtin_amdgcn_readfirstlane(before);
	if (p + 1 < C.nphases && before == 0) {
		return e2_sweep(C, p + 1);
	}
	return true;
}

template <int R, int D, int REAL>
__global__ __launch_bounds__((E2_NC + 1) * 64) void k_engine2(const E2Phase* __restrict__ ph_global, int nphases, uint2* x0, uint2* x1, unsigned tag_base, unsigned* timeout, int flags,
                                                              unsigned long long* stamps, unsigned* diag, uint2* stage) {
	extern __shared__ __attribute__((aligned(1024))) unsigned char e2_smem[];
	unsigned char* ring = e2_smem;                          // R x 16 KiB
	float* xs = (float*)(e2_smem + (size_t)R * E2_SLOT);    // VEC floats: the current phase's input vector
	E2Phase* ph = (E2Phase*)(xs + VEC);                     // E2_MAXP descriptors
	int* cum = (int*)(ph + E2_MAXP);                        // E2_MAXP + 1
	int* leftp = cum + E2_MAXP + 1;                         // [E2_MAXP] consumers that have left phase p
	int* ctl = leftp + E2_MAXP;                             // control words, below
	int* landed = ctl + 0;   // slots landed (monotonic), loader
	int* done = ctl + 1;     // [NC] slots consumed by each consumer (monotonic)
	int* xready = ctl + 5;   // phases whose vector is staged (monotonic): xs holds the input of phase *xready - 1
	int* sweeping = ctl + 6; // a consumer of this workgroup is sweeping granules
	int* gave_up = ctl + 7;
	float* scale_s = (float*)(ctl + 8);
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	for (int i = threadIdx.x; i < nphases; i += (E2_NC + 1) * 64) {
		ph[i] = ph_global[i];
	}
	if (threadIdx.x < 16) {
		ctl[threadIdx.x] = 0;
	}
	for (int i = threadIdx.x; i < E2_MAXP; i += (E2_NC + 1) * 64) {
		leftp[i] = 0;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int c = 0;
		for (int p = 0; p < nphases; ++p) {
			cum[p] = c;
			c += e2_count(ph[p].nslots, blockIdx.x, gridDim.x);
		}
		cum[nphases] = c;
	}
	__syncthreads();
	const int G = cum[nphases];

	if (wave == 0) {
		e2_loader<R, D>(e2_lds_addr(ph), e2_lds_addr(cum), e2_lds_addr(ring), e2_lds_addr(landed), e2_lds_addr(done), e2_lds_addr(sweeping), e2_lds_addr(gave_up), e2_lds_addr(ctl + 9), G, lane,
		                (flags & E2_THIN) != 0);
		if (lane == 0 && diag) {
			diag[blockIdx.x * 8 + 0] = (unsigned)e2_lds_read(e2_lds_addr(ctl + 9)); // polls of a full ring
		}
		return;
	}

	// ------------------------------------------------------------------------------------------ consumers
	const int c = wave - 1;
	E2C C;
	C.xs = xs, C.ph = ph, C.leftp = leftp, C.xready = xready, C.sweeping = sweeping, C.gave_up = gave_up, C.scale_s = scale_s;
	C.x0 = x0, C.x1 = x1, C.tag_base = tag_base, C.nphases = nphases, C.lane = lane, C.noedge = flags & E2_NOEDGE;
	C.stamps = stamps;
	C.stage = (flags & E2_TREE) ? stage : nullptr;
	unsigned waits_landed = 0, waits_vector = 0; // polls spent waiting for the loader / for the phase's vector (diagnostic)
	bool alive = true;
	if (c == 0) {
		alive = e2_sweep(C, 0);
	}
	int p = 0;
	float scale = 0.f;
	int scaled_for = -1;
	for (int g = c; g < G && alive; g += E2_NC) {
		while (g >= cum[p + 1] && alive) {
			alive = e2_leave_phase(C, p);
			++p;
		}
		if (!alive) {
			break;
		}
		if (scaled_for != p) {
			if (!e2_wait_ge(xready, p + 1, gave_up, 7, &waits_vector)) {
				break;
			}
			scale = *scale_s;
			scaled_for = p;
		}
		if (!e2_wait_ge(landed, g + 1, gave_up, 8, &waits_landed)) {
			break;
		}
		const unsigned k = (unsigned)(g - cum[p]);
		const unsigned s = blockIdx.x + gridDim.x * k; // slot of the phase: tasks 2 s, 2 s + 1
		const unsigned ntasks = 2u * ph[p].nslots;
		const u32x4* slot = (const u32x4*)(ring + (size_t)(g % R) * E2_SLOT);
		uint2* xout = (p & 1) ? x0 : x1;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			u32x4 tile[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				tile[u] = slot[(h * 8 + u) * 64 + lane];
			}
			const unsigned t = 2 * s + h;
			const float tv = tile_value<REAL>(tile, xs, lane);
			const float v = wave_sum(tv) * scale * xs[(t * 7) % VEC];
			const unsigned oi = e2_out_index(t, ntasks);
			if (lane == 0 && oi < (unsigned)VEC) {
				e2_publish(xout + oi, v + (float)(t % 13), tag_base + 4u + (unsigned)p);
			}
		}
		if (lane == 0) {
			__hip_atomic_store(&done[c], g / E2_NC + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // (the slot's reads have returned: v depends on them)
		}
	}
	while (p < nphases && alive) { // phases after this consumer's last slot
		alive = e2_leave_phase(C, p);
		++p;
	}
	if (lane == 0 && *gave_up) {
		*timeout = (unsigned)*gave_up;
	}
	if (lane == 0 && diag) {
		diag[blockIdx.x * 8 + 1 + c] = waits_landed;
		diag[blockIdx.x * 8 + 4 + c] = waits_vector;
	}
}

// ---- the same chain, one plain launch per phase, over the same granule buffers (the checker, and a launch-structure figure)
template <int REAL>
__global__ __launch_bounds__(BLOCK) void k_ref_phase(const void* w, unsigned ntasks, const uint2* xin, uint2* xout, unsigned tag) {
	__shared__ float red[NW];
	__shared__ float xs[VEC];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	for (int i = threadIdx.x; i < VEC; i += BLOCK) {
		xs[i] = __builtin_bit_cast(float, xin[i].x);
	}
	__syncthreads();
	const f32x4 a = ((const f32x4*)xs)[threadIdx.x], b = ((const f32x4*)xs)[threadIdx.x + BLOCK];
	float ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
	ss = wave_sum(ss);
	if (lane == 0) {
		red[wave] = ss;
	}
	__syncthreads();
	float tot = 0.f;
#pragma unroll
	for (int i = 0; i < NW; ++i) {
		tot += red[i];
	}
	const float scale = 1.0f / sqrtf(tot / VEC + 1e-5f);
	for (unsigned t = blockIdx.x * NW + wave; t < ntasks; t += gridDim.x * NW) {
		u32x4 tile[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			tile[u] = __builtin_nontemporal_load((gptr16)w + (size_t)t * 512 + u * 64 + lane);
		}
		const float tv = tile_value<REAL>(tile, xs, lane);
		const float v = wave_sum(tv) * scale * xs[(t * 7) % VEC];
		const unsigned oi = e2_out_index(t, ntasks);
		if (lane == 0 && oi < (unsigned)VEC) {
			const uint2 d = {__builtin_bit_cast(unsigned, v + (float)(t % 13)), tag};
			xout[oi] = d;
		}
	}
}

__global__ void k_e2_init(uint2* x0, uint2* x1, unsigned tag_base) { // the vectors "phases -1 and -2" left behind
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < VEC) {
		x0[i] = make_uint2(__builtin_bit_cast(unsigned, 0.001f * (i % 97) + 0.5f), tag_base + 4u - 1u);
		x1[i] = make_uint2(__builtin_bit_cast(unsigned, 0.25f + 0.002f * (i % 53)), tag_base + 4u - 2u);
	}
}

struct E2Setup {
	std::vector<void*> w;
	std::vector<E2Phase> hp;
	E2Phase* dp = nullptr;
	uint2* x[2] = {nullptr, nullptr};
	unsigned* timeout = nullptr;
	unsigned long long* stamps = nullptr; // [phase][256][4]
	unsigned* diag = nullptr;             // [256][8]
	uint2* stage = nullptr;               // [8][2][VEC]
	hipStream_t s = nullptr;
	int total = 0;
};

static const size_t e2_sizes[4] = {25165824, 16777216, 117440512, 58720256};

static void e2_setup(E2Setup& S, int n_layers) {
	S.total = n_layers * 4;
	CK(hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking));
	S.w.resize(S.total);
	S.hp.resize(S.total);
	for (int i = 0; i < S.total; ++i) {
		CK(hipMalloc(&S.w[i], e2_sizes[i % 4] + 65536));
		CK(hipMemset(S.w[i], 0x11 + i / 4 + i % 4, e2_sizes[i % 4] + 65536));
		S.hp[i].w = S.w[i];
		S.hp[i].nslots = (unsigned)(e2_sizes[i % 4] / E2_SLOT);
		S.hp[i].pad = 0;
	}
	CK(hipMalloc(&S.dp, sizeof(E2Phase) * S.total));
	CK(hipMemcpy(S.dp, S.hp.data(), sizeof(E2Phase) * S.total, hipMemcpyHostToDevice));
	CK(hipMalloc(&S.x[0], VEC * 8 + 65536));
	CK(hipMalloc(&S.x[1], VEC * 8 + 65536));
	CK(hipMalloc(&S.timeout, 4));
	CK(hipMemset(S.timeout, 0, 4));
	CK(hipMalloc(&S.stamps, (size_t)S.total * 256 * 4 * 8));
	CK(hipMemset(S.stamps, 0, (size_t)S.total * 256 * 4 * 8));
	CK(hipMalloc(&S.diag, 256 * 8 * 4));
	CK(hipMemset(S.diag, 0, 256 * 8 * 4));
	CK(hipMalloc(&S.stage, (size_t)8 * 2 * VEC * 8 + 65536));
	CK(hipMemset(S.stage, 0, (size_t)8 * 2 * VEC * 8));
}
static void e2_teardown(E2Setup& S) {
	for (void* p : S.w) {
		CK(hipFree(p));
	}
	CK(hipFree(S.dp));
	CK(hipFree(S.x[0]));
	CK(hipFree(S.x[1]));
	CK(hipFree(S.timeout));
	CK(hipFree(S.stamps));
	CK(hipFree(S.diag));
	CK(hipFree(S.stage));
	CK(hipStreamDestroy(S.s));
}
static double e2_checksum(E2Setup& S, std::vector<float>* keep) {
	std::vector<uint2> xf(VEC);
	CK(hipMemcpy(xf.data(), S.x[S.total & 1], VEC * 8, hipMemcpyDeviceToHost));
	double cs = 0;
	if (keep) {
		keep->resize(VEC);
	}
	for (int i = 0; i < VEC; ++i) {
		float v;
		memcpy(&v, &xf[i].x, 4);
		cs += v * (1 + i % 5);
		if (keep) {
			(*keep)[i] = v;
		}
	}
	return cs;
}

static int g_e2_flags = E2_THIN;
static int g_e2_report = 0; // 1: the kernel stamps every sweep and counts its polls; e2_run prints a digest of the last launch
static std::vector<float> g_e2_ref; // final vector of the launch-per-phase chain (the checker)

// launch-per-phase over the granule buffers: us/layer (plain launches on one stream), fills g_e2_ref
extern "C" double exp_engine2_ref(int real, int n_layers, int iters, double* checksum) {
	E2Setup S;
	e2_setup(S, n_layers);
	unsigned tag_base = 1u << 12;
	auto run = [&]() {
		hipLaunchKernelGGL(k_e2_init, dim3((VEC + 255) / 256), dim3(256), 0, S.s, S.x[0], S.x[1], tag_base);
		for (int p = 0; p < S.total; ++p) {
			const uint2* xin = (p & 1) ? S.x[1] : S.x[0];
			uint2* xout = (p & 1) ? S.x[0] : S.x[1];
			if (real) {
				hipLaunchKernelGGL(k_ref_phase<1>, dim3(256), dim3(BLOCK), 0, S.s, S.hp[p].w, 2 * S.hp[p].nslots, xin, xout, tag_base + 4u + (unsigned)p);
			} else {
				hipLaunchKernelGGL(k_ref_phase<0>, dim3(256), dim3(BLOCK), 0, S.s, S.hp[p].w, 2 * S.hp[p].nslots, xin, xout, tag_base + 4u + (unsigned)p);
			}
		}
		tag_base += 1u << 12;
	};
	run();
	CK(hipGetLastError());
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, S.s));
	for (int i = 0; i < iters; ++i) {
		run();
	}
	CK(hipEventRecord(e1, S.s));
	CK(hipDeviceSynchronize());
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	*checksum = e2_checksum(S, &g_e2_ref);
	e2_teardown(S);
	return (double)ms * 1e3 / ((double)iters * n_layers);
}

template <int R, int D, int REAL>
static double e2_run(int n_layers, int iters, double* checksum, int* mismatches) {
	E2Setup S;
	e2_setup(S, n_layers);
	if (S.total > E2_MAXP) {
		fprintf(stderr, "exp_engine2: at most %d phases\n", E2_MAXP);
		return -1;
	}
	const size_t lds = (size_t)R * E2_SLOT + VEC * 4 + sizeof(E2Phase) * E2_MAXP + 4 * (E2_MAXP + 1) + 4 * E2_MAXP + 64;
	auto kern = k_engine2<R, D, REAL>;
	CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	unsigned tag_base = 1u << 12;
	auto run = [&]() {
		hipLaunchKernelGGL(k_e2_init, dim3((VEC + 255) / 256), dim3(256), 0, S.s, S.x[0], S.x[1], tag_base);
		hipLaunchKernelGGL(kern, dim3(256), dim3((E2_NC + 1) * 64), lds, S.s, (const E2Phase*)S.dp, S.total, S.x[0], S.x[1], tag_base, S.timeout, g_e2_flags,
		                   g_e2_report ? S.stamps : (unsigned long long*)nullptr, g_e2_report ? S.diag : (unsigned*)nullptr, S.stage);
		tag_base += 1u << 12;
	};
	run();
	CK(hipGetLastError());
	CK(hipDeviceSynchronize());
	unsigned to = 0;
	CK(hipMemcpy(&to, S.timeout, 4, hipMemcpyDeviceToHost));
	double us = -1;
	if (to) {
		printf("  !! a bounded spin gave up in the first launch (R=%d D=%d, code %u): not timed\n", R, D, to);
	} else {
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0, S.s));
		for (int i = 0; i < iters; ++i) {
			run();
		}
		CK(hipEventRecord(e1, S.s));
		CK(hipDeviceSynchronize());
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		us = (double)ms * 1e3 / ((double)iters * n_layers);
		CK(hipMemcpy(&to, S.timeout, 4, hipMemcpyDeviceToHost));
		if (to) {
			printf("  !! a bounded spin gave up during the timed launches (R=%d D=%d, code %u)\n", R, D, to);
		}
	}
	std::vector<float> got;
	*checksum = e2_checksum(S, &got);
	int bad = 0;
	if (g_e2_ref.size() == got.size()) {
		for (size_t i = 0; i < got.size(); ++i) {
			bad += memcmp(&got[i], &g_e2_ref[i], 4) != 0;
		}
	} else {
		bad = -1;
	}
	*mismatches = bad;
	if (g_e2_report) {
		// digest of the LAST launch: per phase kind (qkv / wo / ffn_up / ffn_down inputs), over layers >= 1 and all workgroups:
		// how long a sweep polls before every tag is there, and how long it then takes to stage the vector; who waits for whom
		std::vector<unsigned long long> st((size_t)S.total * 256 * 4);
		std::vector<unsigned> dg(256 * 8);
		CK(hipMemcpy(st.data(), S.stamps, st.size() * 8, hipMemcpyDeviceToHost));
		CK(hipMemcpy(dg.data(), S.diag, dg.size() * 4, hipMemcpyDeviceToHost));
		static const char* kind[4] = {"qkv", "wo", "ffn_up", "ffn_down"};
		for (int k = 0; k < 4; ++k) {
			double poll = 0, stage = 0, rounds = 0, span = 0;
			int n = 0;
			for (int p = 4 + k; p < S.total; p += 4) {
				unsigned long long first_begin = ~0ull, last_ready = 0;
				for (int b = 0; b < 256; ++b) {
					const unsigned long long* o = &st[((size_t)p * 256 + b) * 4];
					if (!o[2]) {
						continue;
					}
					poll += (double)(o[1] - o[0]) * 0.01, stage += (double)(o[2] - o[1]) * 0.01, rounds += (double)o[3];
					first_begin = o[0] < first_begin ? o[0] : first_begin, last_ready = o[2] > last_ready ? o[2] : last_ready;
					++n;
				}
				span += (double)(last_ready - first_begin) * 0.01;
			}
			if (n) {
				printf("    edge into %-8s: sweep polls %5.2f us (%4.1f rounds), then stages in %5.2f us; first sweep begun -> last vector staged %5.2f us\n", kind[k], poll / n,
				       rounds / n, stage / n, span / (S.total / 4 - 1));
			}
		}
		double full = 0, wl = 0, wv = 0;
		for (int b = 0; b < 256; ++b) {
			full += dg[b * 8];
			wl += dg[b * 8 + 1] + dg[b * 8 + 2] + dg[b * 8 + 3];
			wv += dg[b * 8 + 4] + dg[b * 8 + 5] + dg[b * 8 + 6];
		}
		printf("    polls per workgroup over the launch: loader on a full ring %.0f, consumers on the loader %.0f, consumers on the vector %.0f\n", full / 256, wl / 256, wv / 256);
	}
	fflush(stdout);
	e2_teardown(S);
	return us;
}

extern "C" void exp_engine2_knobs(int noedge, int thin, int report, int tree) {
	g_e2_flags = (noedge ? E2_NOEDGE : 0) | (thin ? E2_THIN : 0) | (tree ? E2_TREE : 0);
	g_e2_report = report;
}

// config = R * 10 + D
extern "C" double exp_engine2(int config, int real, int n_layers, int iters, double* checksum, int* mismatches) {
#define E2(r, d)                                                                  \
	if (config == r * 10 + d) {                                                   \
		return real ? e2_run<r, d, 1>(n_layers, iters, checksum, mismatches) : e2_run<r, d, 0>(n_layers, iters, checksum, mismatches); \
	}
	E2(8, 3)
	E2(8, 4)
	E2(8, 2)
	E2(6, 3)
	E2(4, 2)
#undef E2
	fprintf(stderr, "exp_engine2: config %d not built\n", config);
	return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Loader study: how fast can LDS-DMA alone stream a layer's weights, with NL loader waves per workgroup (one workgroup per CU),
// D fills of 16 KiB in flight per wave, slots assigned in contiguous runs of RUN slots per workgroup (RUN = 0: one contiguous
// block per workgroup; RUN = 1: fully interleaved), non-temporal or default cache policy?  Nobody consumes: a slot is free again as
// soon as it has landed.  (The engine cannot beat its loader.)
template <int NL, int D, int NT>
__global__ __launch_bounds__(NL * 64) void k_loader_only(const E2Phase* __restrict__ ph, int nphases, int run, unsigned* sink) {
	extern __shared__ __attribute__((aligned(1024))) unsigned char e2_smem[];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const unsigned a_ring = e2_lds_addr(e2_smem) + (unsigned)wave * (unsigned)(D * E2_SLOT); // each wave fills its own D slots
	unsigned n = 0;
	for (int p = 0; p < nphases; ++p) {
		const unsigned char* wbase = (const unsigned char*)ph[p].w;
		const unsigned nslots = ph[p].nslots, per = (nslots + gridDim.x - 1) / gridDim.x; // slots per workgroup
		for (unsigned k = wave; k < per; k += NL, ++n) {
			// k-th slot of this workgroup in the phase
			unsigned s;
			if (run <= 0) {
				s = blockIdx.x * per + k;
			} else {
				s = ((k / run) * gridDim.x + blockIdx.x) * run + k % run;
			}
			if (s >= nslots) {
				continue;
			}
			const unsigned char* src = wbase + (size_t)s * E2_SLOT + lane * 16;
			const unsigned slot = a_ring + (n % D) * E2_SLOT;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const __attribute__((address_space(1))) void* gp = (const __attribute__((address_space(1))) void*)(src + q * 4096);
				const e2_lds_t lp = (e2_lds_t)(size_t)(slot + q * 4096);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 0, NT ? 2 : 0);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, NT ? 2 : 0);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, NT ? 2 : 0);
				__builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, NT ? 2 : 0);
			}
			if (D == 1) {
				e2_vmcnt<0>();
			} else {
				e2_vmcnt<16 * (D - 1) < 63 ? 16 * (D - 1) : 63>();
			}
		}
	}
	e2_vmcnt<0>();
	if (sink && e2_lds_read(a_ring + lane * 4) == 0x9e3779b9) {
		*sink = 1;
	}
}

// the same stream through registers (global_load_dwordx4 nt, W waves per workgroup, 16 KiB in flight per wave, nothing kept)
template <int W>
__global__ __launch_bounds__(W * 64) void k_reg_stream(const E2Phase* __restrict__ ph, int nphases, unsigned* sink) {
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	unsigned acc = 0;
	for (int p = 0; p < nphases; ++p) {
		const gptr16 w = (gptr16)ph[p].w;
		const unsigned nslots = ph[p].nslots;
		for (unsigned s = blockIdx.x * W + wave; s < nslots; s += gridDim.x * W) {
			u32x4 t[16];
#pragma unroll
			for (int u = 0; u < 16; ++u) {
				t[u] = __builtin_nontemporal_load(w + (size_t)s * 1024 + u * 64 + lane);
			}
#pragma unroll
			for (int u = 0; u < 16; ++u) {
				acc += t[u][0] ^ t[u][3];
			}
		}
	}
	if (acc == 0x9e3779b9u) {
		*sink = acc;
	}
}

// config = NL * 100 + D * 10 + NT;  run: slots per contiguous run (0 = one block per workgroup); NL = 9: the register stream (8 waves)
extern "C" double exp_loader_only(int config, int run, int n_layers, int iters) {
	E2Setup S;
	e2_setup(S, n_layers);
	auto go = [&](auto kern, int threads, size_t lds, bool has_run) {
		if (lds > 48 * 1024) {
			CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		}
		(void)has_run;
		hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, S.s, (const E2Phase*)S.dp, S.total, run, S.timeout);
	};
	auto timed = [&](auto launch) {
		launch();
		CK(hipGetLastError());
		CK(hipDeviceSynchronize());
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0, S.s));
		for (int i = 0; i < iters; ++i) {
			launch();
		}
		CK(hipEventRecord(e1, S.s));
		CK(hipDeviceSynchronize());
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		return (double)ms * 1e3 / ((double)iters * n_layers);
	};
	double us = -1;
#define LO(nl, d, nt)                                                                                                     \
	if (config == nl * 100 + d * 10 + nt) {                                                                               \
		us = timed([&]() { go(k_loader_only<nl, d, nt>, nl * 64, (size_t)nl * d * E2_SLOT, true); });                      \
	}
	LO(1, 2, 1) LO(1, 3, 1) LO(1, 4, 1) LO(1, 3, 0) LO(2, 2, 1) LO(2, 3, 1) LO(2, 4, 1) LO(4, 2, 1) LO(4, 1, 1) LO(2, 3, 0) LO(3, 3, 1)
#undef LO
	if (config == 900) {
		us = timed([&]() { hipLaunchKernelGGL(k_reg_stream<8>, dim3(256), dim3(512), 0, S.s, (const E2Phase*)S.dp, S.total, S.timeout); });
	}
	if (config == 901) {
		us = timed([&]() { hipLaunchKernelGGL(k_reg_stream<4>, dim3(512), dim3(256), 0, S.s, (const E2Phase*)S.dp, S.total, S.timeout); });
	}
	e2_teardown(S);
	return us;
}
