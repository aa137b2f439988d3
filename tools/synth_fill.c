/*
 * synth_fill.c -- fast multi-threaded filler for synthetic .calm weights (tooling, not hot path).
 *
 * calm_amd/calmfile.py samples weight CODES through a 65536-entry inverse-CDF table; numpy does that
 * at ~150 MB/s under the GIL, which makes a 7 GB Mistral-shaped model take a minute.  This does the
 * same table walk with OpenMP and a per-chunk splitmix64/xoshiro stream: several GB/s.
 *
 *   kind 0: out is uint8[n],  out[i] = lut8[u16]                       (fp8 e5m2 codes)
 *   kind 1: out is uint16[n], out[i] = lut16[u16]                      (fp16 bit patterns)
 *   kind 2: out is uint32[n], gf4 words: scale = lut8[u16], 8 random 3-bit codes, one forced to 0
 *
 * build: gcc -O3 -fopenmp -fPIC -shared -o tools/libsynth_fill.so tools/synth_fill.c
 */
#include <omp.h>
#include <stddef.h>
#include <stdint.h>

static inline uint64_t splitmix64(uint64_t* s) {
	uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}

void synth_fill(void* out, size_t n, int kind, const void* lut, uint64_t seed) {
	const size_t chunk = 1 << 20;
	const size_t nchunks = (n + chunk - 1) / chunk;
	long c;
	/* at most 32 threads, and no more than there are chunks: with the runtime's default team (256 on the GPU hosts) the 291 small
	 * parallel regions of a model spent 28 s waking and spinning threads for 7 GB of table look-ups (0.3 s of upload behind them) */
	int team = omp_get_max_threads();
	team = team > 32 ? 32 : team;
	team = (size_t)team > nchunks ? (int)nchunks : team;
	team = team < 1 ? 1 : team;
#pragma omp parallel for schedule(dynamic, 4) num_threads(team)
	for (c = 0; c < (long)nchunks; ++c) {
		uint64_t s = seed * 0x2545f4914f6cdd1dull + (uint64_t)c * 0x9e3779b97f4a7c15ull + 1;
		size_t a = (size_t)c * chunk, b = a + chunk < n ? a + chunk : n;
		if (kind == 0) {
			uint8_t* o = (uint8_t*)out;
			const uint8_t* l = (const uint8_t*)lut;
			size_t i = a;
			for (; i + 4 <= b; i += 4) {
				uint64_t r = splitmix64(&s);
				o[i] = l[r & 0xffff], o[i + 1] = l[(r >> 16) & 0xffff], o[i + 2] = l[(r >> 32) & 0xffff], o[i + 3] = l[r >> 48];
			}
			for (; i < b; ++i) {
				o[i] = l[splitmix64(&s) & 0xffff];
			}
		} else if (kind == 1) {
			uint16_t* o = (uint16_t*)out;
			const uint16_t* l = (const uint16_t*)lut;
			size_t i = a;
			for (; i + 4 <= b; i += 4) {
				uint64_t r = splitmix64(&s);
				o[i] = l[r & 0xffff], o[i + 1] = l[(r >> 16) & 0xffff], o[i + 2] = l[(r >> 32) & 0xffff], o[i + 3] = l[r >> 48];
			}
			for (; i < b; ++i) {
				o[i] = l[splitmix64(&s) & 0xffff];
			}
		} else {
			uint32_t* o = (uint32_t*)out;
			const uint8_t* l = (const uint8_t*)lut;
			for (size_t i = a; i < b; ++i) {
				uint64_t r = splitmix64(&s);
				uint32_t w = (uint32_t)(r >> 32) & 0xffffff00u;
				uint32_t k = (uint32_t)(r & 7) * 3 + 8;
				w &= ~(7u << k);
				o[i] = w | l[(r >> 8) & 0xffff];
			}
		}
	}
}
