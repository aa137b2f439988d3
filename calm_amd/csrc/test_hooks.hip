// test_hooks.hip -- libcalm_hip_test.so: unit-level entry points for tests/ and tools/ (include/calm_hip_test.h).
//
// NOT part of the drop-in library: libcalm_hip.so exports the backend ABI of include/calm_hip.h and nothing else.  This
// translation unit compiles the product's kernels and launch helpers a second time (it includes the product source, whose
// helpers are file-local) and adds hooks that run single kernels on caller-provided buffers, reach into a prepared
// transformer's KV cache through the fields struct Transformer shows, and a streaming-read micro-benchmark.
#include "infer_hip.hip"

#include "../../include/calm_hip_test.h"

// ================================================================ test hooks ==================

namespace {

template <class F>
void by_dbits(int dbits, F f) {
	switch (dbits) {
	case 16:
		f(std::integral_constant<int, 16>());
		break;
	case 8:
		f(std::integral_constant<int, 8>());
		break;
	case 4:
		f(std::integral_constant<int, 4>());
		break;
	default:
		CALM_REQUIRE(false, "dbits must be 4, 8 or 16");
	}
}

} // namespace

extern "C" void calm_hip_test_matvec(int dbits, const void* w, const float* x, float* out, int n, int d) {
	init_hip();
	CALM_REQUIRE(n % (128 / dbits) == 0 && d % 4 == 0, "n must be a multiple of 128/dbits and d of 4");
	size_t wbytes = (size_t)n * d * dbits / 8;
	void* dw = upload_hip((void*)w, wbytes);
	float* dx = (float*)upload_hip((void*)x, n * sizeof(float));
	float* dout = (float*)dev_alloc(d * sizeof(float));
	dev_zero(dout, d * sizeof(float)); // (k_attn_out adds into it) -- on the decode stream, like the launch behind it
	by_dbits(dbits, [&](auto DBT) {
		constexpr int DB = decltype(DBT)::value;
		by_bool(stage_v4(n, WG_THREADS), [&](auto V4) {
			by_bool(rows_full<DB>(n), [&](auto FULL) {
				auto k = k_attn_out<DB, decltype(V4)::value ? 4 : 8, decltype(FULL)::value, false, false>;
				allow_lds(k, lds_bytes<DB>(n));
				hipLaunchKernelGGL(k, dim3(pick_blocks_wg(d / KShape<DB, KS_ATTN_OUT>::NR)), dim3(WG_THREADS), lds_bytes<DB>(n), g_stream, dout, dx, dw, d, n, (const float*)nullptr, (float*)nullptr, 0);
			});
		});
	});
	HIP_CHECK(hipGetLastError());
	download_hip(out, dout, d * sizeof(float));
	free_hip(dw), free_hip(dx), free_hip(dout);
}

extern "C" void calm_hip_test_norm_matvec(int dbits, const void* w, const float* x, const float* nw, float* out, int n, int d, float eps, int ln) {
	init_hip();
	CALM_REQUIRE(n % (128 / dbits) == 0, "n must be a multiple of 128/dbits");
	size_t wbytes = (size_t)n * d * dbits / 8;
	void* dw = upload_hip((void*)w, wbytes);
	float* dx = (float*)upload_hip((void*)x, n * sizeof(float));
	float* dnw = (float*)upload_hip((void*)nw, n * sizeof(float));
	float* dout = (float*)dev_alloc(d * sizeof(float));
	dev_fill(dout, 0xff, d * sizeof(float)); // poison (NaN): a row the kernel skipped cannot look like an answer
	by_dbits(dbits, [&](auto DBT) {
		constexpr int DB = decltype(DBT)::value;
		int ntasks = (d + KShape<DB, KS_OUTPUT>::NR - 1) / KShape<DB, KS_OUTPUT>::NR;
		by_bool(stage_v4(n, WG_THREADS), [&](auto V4) {
			by_bool(rows_full<DB>(n), [&](auto FULL) {
				auto k = k_output<DB, decltype(V4)::value ? 4 : 8, decltype(FULL)::value>;
				allow_lds(k, lds_bytes<DB>(n));
				hipLaunchKernelGGL(k, dim3(pick_blocks_wg(ntasks)), dim3(WG_THREADS), lds_bytes<DB>(n), g_stream, dout, dx, dnw, dw, n, d, eps, ln, 0);
			});
		});
	});
	HIP_CHECK(hipGetLastError());
	download_hip(out, dout, d * sizeof(float));
	free_hip(dw), free_hip(dx), free_hip(dnw), free_hip(dout);
}

extern "C" void calm_hip_test_attn(const float* q, const uint16_t* kcache, const uint16_t* vcache, float* out, int n_heads, int n_kv_heads, int head_dim,
                                   int seq_len, int kv_len, int n_split) {
	init_hip();
	CALM_REQUIRE(head_dim % 8 == 0 && n_heads % n_kv_heads == 0 && n_split >= 1 && n_split <= MAX_SPLIT, "bad attention test shape");
	int kv_dim = n_kv_heads * head_dim, q_dim = n_heads * head_dim;
	// oracle layout [seq_len][kv_dim] -> backend layout [kv_head][seq_len][head_dim]
	std::vector<uint16_t> kk((size_t)seq_len * kv_dim), vv((size_t)seq_len * kv_dim);
	for (int t = 0; t < seq_len; ++t) {
		for (int h = 0; h < n_kv_heads; ++h) {
			for (int d = 0; d < head_dim; ++d) {
				kk[((size_t)h * seq_len + t) * head_dim + d] = kcache[(size_t)t * kv_dim + h * head_dim + d];
				vv[((size_t)h * seq_len + t) * head_dim + d] = vcache[(size_t)t * kv_dim + h * head_dim + d];
			}
		}
	}
	Ctx c;
	c.head_dim = head_dim, c.n_heads = n_heads, c.n_kv_heads = n_kv_heads, c.kv_mul = n_heads / n_kv_heads, c.seq_len = seq_len;
	c.kv_layer_bytes = kk.size() * 2;
	c.kc = upload_hip(kk.data(), kk.size() * 2);
	c.vc = upload_hip(vv.data(), vv.size() * 2);
	if (attn_has_vt(head_dim) && seq_len % 64 == 0) { // the transposed copy k_attn_vt reads (kernels.hip.h attn_vt_offset)
		std::vector<uint16_t> tt((size_t)seq_len * kv_dim);
		for (int t = 0; t < seq_len; ++t) {
			for (int hd = 0; hd < kv_dim; ++hd) {
				tt[attn_vt_offset(hd, t, head_dim, seq_len, 2)] = vcache[(size_t)t * kv_dim + hd];
			}
		}
		c.vt = upload_hip(tt.data(), tt.size() * 2);
	}
	c.q = (float*)upload_hip((void*)q, q_dim * sizeof(float));
	c.att = (float*)dev_alloc(q_dim * sizeof(float));
	dev_fill(c.att, 0xff, q_dim * sizeof(float)); // poison (NaN)
	c.partial = (float*)dev_alloc((size_t)n_heads * MAX_SPLIT * (head_dim + 4) * sizeof(float));
	TokState ts = {};
	ts.kv_len = kv_len;
	c.ts = (TokState*)upload_hip(&ts, sizeof(ts));
	c.lpr = 4;
	while (c.lpr * 8 < head_dim) {
		c.lpr *= 2;
	}
	c.attn_chunk = (kv_len + n_split - 1) / n_split;
	launch_attn<16>(&c, 0, n_split);
	HIP_CHECK(hipGetLastError());
	download_hip(out, c.att, q_dim * sizeof(float));
	free_hip(c.kc), free_hip(c.vc), free_hip(c.q), free_hip(c.att), free_hip(c.partial), free_hip(c.ts);
	if (c.vt) {
		free_hip(c.vt);
	}
}

extern "C" int calm_hip_test_argmax(const float* logits, int n) {
	init_hip();
	float* dl = (float*)upload_hip((void*)logits, n * sizeof(float));
	int* dn = (int*)dev_alloc(2 * sizeof(int));
	dev_zero(dn, 2 * sizeof(int));
	hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, g_stream, dl, n, dn, (int*)nullptr, dn + 1);
	HIP_CHECK(hipGetLastError());
	int r = -2;
	download_hip(&r, dn, sizeof(int));
	free_hip(dl), free_hip(dn);
	return r;
}

namespace {
__global__ void k_test_pf_pack(void* out, const float* X, int K) { // row-major fp32 -> the fragment-major hi + lo matrix (prefill.hip.h: pf_unit)
	const int t = blockIdx.x;
	for (int i = threadIdx.x; i < K / 8; i += blockDim.x) {
		float v[8];
		for (int e = 0; e < 8; ++e) {
			v[e] = X[(size_t)t * K + 8 * i + e];
		}
		pf_store8(out, t, 8 * i, pf_steps(K), v);
	}
}
} // namespace

extern "C" void calm_hip_test_pf_gemm(int dbits, const void* w, const float* x, float* out, int M, int K, int nb, int form) {
	init_hip();
	CALM_REQUIRE(K % 32 == 0 && M % 4 == 0 && nb > 0, "K must be a multiple of 32 and M of 4");
	const int cols = (nb + 63) / 64;
	const size_t wbytes = (size_t)M * K * dbits / 8;
	void* dw = upload_hip((void*)w, wbytes);
	float* dx = (float*)upload_hip((void*)x, (size_t)nb * K * sizeof(float));
	const size_t fbytes = (size_t)((nb + 127) / 128 * 128) * pf_steps(K) * 64 * sizeof(float); // (whole 128-token columns: the big form's)
	void* dxf = dev_alloc(fbytes);
	HIP_CHECK(hipMemsetAsync(dxf, 0, fbytes, g_stream));
	float* dout = (float*)dev_alloc((size_t)nb * M * sizeof(float));
	HIP_CHECK(hipMemsetAsync(dout, 0, (size_t)nb * M * sizeof(float), g_stream));
	hipLaunchKernelGGL(k_test_pf_pack, dim3(nb), dim3(256), 0, g_stream, dxf, dx, K);
	PfGemmArgs a;
	memset(&a, 0, sizeof(a));
	a.xin = (const float4*)dxf, a.w0 = dw, a.K = K, a.M = M, a.nb = nb, a.out = dout, a.clip = 3.4e38f;
	a.ncols = cols, a.ksplit = 1;
	constexpr int UNITS = PfWide<PF_EPI_STORE>::UNITS;
	const int nx = (M + UNITS - 1) / UNITS, tiles = 8 * ((nx + 7) / 8) * cols;
	const int big_ks = form >= 90 ? form - 90 : (form == 9 ? 1 : 0); // 9: k_pf_gemm_big; 92 .. 98: with K cut into 2 .. 8 ranges
	size_t big_tiles = 0;
	if (big_ks > 1) {
		CALM_REQUIRE(big_ks <= 8 && pf_steps(K) >= big_ks, "K ranges: 2..8, at least one step each");
		big_tiles = (size_t)8 * (((M + 511) / 512 + 7) / 8) * ((nb + 127) / 128);
		a.ksplit = big_ks;
		a.partial = (float*)dev_alloc(big_tiles * big_ks * 65536 * sizeof(float));
		a.tile_count = (unsigned*)dev_alloc(big_tiles * sizeof(unsigned));
		HIP_CHECK(hipMemsetAsync(a.tile_count, 0, big_tiles * sizeof(unsigned), g_stream));
	}
	if (form >= 2 && !big_ks) {
		CALM_REQUIRE(form <= 8 && pf_steps(K) >= form, "K ranges: 2..8, at least one step each");
		a.ksplit = form;
		a.partial = (float*)dev_alloc((size_t)tiles * form * 16384 * sizeof(float));
		a.tile_count = (unsigned*)dev_alloc((size_t)tiles * sizeof(unsigned));
		HIP_CHECK(hipMemsetAsync(a.tile_count, 0, (size_t)tiles * sizeof(unsigned), g_stream));
	}
	by_dbits(dbits, [&](auto DBT) {
		constexpr int DB = decltype(DBT)::value;
		if (big_ks) { // k_pf_gemm_big (fp8 / gf4)
			if constexpr (DB != 16) {
				a.ncols = (nb + 127) / 128;
				auto kern = k_pf_gemm_big<DB, PF_EPI_STORE>;
				allow_lds(kern, PfBigA<DB>::LDS_BYTES);
				hipLaunchKernelGGL(kern, dim3(pf_wide_grid((M + 511) / 512, a.ncols, big_ks)), dim3(512), PfBigA<DB>::LDS_BYTES, g_stream, a);
			} else {
				CALM_REQUIRE(false, "the big form takes fp8 / gf4 weights");
			}
		} else if (form >= 1) {
			auto kern = k_pf_gemm_wide<DB, 16, PF_EPI_STORE, 1>;
			allow_lds(kern, PfWideA<DB>::LDS_BYTES);
			hipLaunchKernelGGL(kern, dim3(pf_wide_grid(nx, cols, a.ksplit)), dim3(256), PfWideA<DB>::LDS_BYTES, g_stream, a);
		} else if (form == 0) {
			hipLaunchKernelGGL((k_pf_gemm<DB, 16, PF_EPI_STORE, 2>), dim3((M + 63) / 64, cols), dim3(256), 0, g_stream, a);
		} else if (form == -1) {
			hipLaunchKernelGGL((k_pf_gemm<DB, 16, PF_EPI_STORE, 1>), dim3((M + 31) / 32, cols), dim3(256), 0, g_stream, a);
		} else {
			hipLaunchKernelGGL((k_pf_gemm<DB, 16, PF_EPI_STORE, 3>), dim3((M + 95) / 96, cols), dim3(256), 0, g_stream, a);
		}
	});
	HIP_CHECK(hipGetLastError());
	download_hip(out, dout, (size_t)nb * M * sizeof(float));
	if (big_ks > 1) {
		std::vector<unsigned> cnt(big_tiles);
		download_hip(cnt.data(), a.tile_count, big_tiles * sizeof(unsigned));
		unsigned left = 0;
		for (unsigned v : cnt) {
			left |= v;
		}
		CALM_REQUIRE(left == 0, "a tile counter was not reset");
		free_hip(a.partial), free_hip(a.tile_count);
	}
	if (form >= 2 && !big_ks) {
		unsigned left = 0; // every tile's counter is back at zero
		std::vector<unsigned> cnt(tiles);
		download_hip(cnt.data(), a.tile_count, (size_t)tiles * sizeof(unsigned));
		for (unsigned v : cnt) {
			left |= v;
		}
		CALM_REQUIRE(left == 0, "a tile counter was not reset");
		free_hip(a.partial), free_hip(a.tile_count);
	}
	free_hip(dw), free_hip(dx), free_hip(dxf), free_hip(dout);
}

extern "C" int calm_hip_test_sample(const float* logits, int n, float temperature, float minp, unsigned long long* rng_state) {
	init_hip();
	float* dl = (float*)upload_hip((void*)logits, n * sizeof(float));
	int* dn = (int*)dev_alloc(2 * sizeof(int));
	dev_zero(dn, 2 * sizeof(int));
	SampleState st;
	st.rng = *rng_state, st.temperature = temperature, st.cutoff_offset = logf(minp) * temperature;
	SampleState* ds = (SampleState*)upload_hip(&st, sizeof(st));
	hipLaunchKernelGGL(k_sample_minp, dim3(1), dim3(1024), 0, g_stream, dl, n, dn, (int*)nullptr, dn + 1, ds);
	HIP_CHECK(hipGetLastError());
	int r = -2;
	download_hip(&r, dn, sizeof(int));
	download_hip(&st, ds, sizeof(st));
	*rng_state = st.rng;
	free_hip(dl), free_hip(dn), free_hip(ds);
	return r;
}

namespace {

// the backend's private cache layout, from what struct Transformer shows: [layer][kv_head][seq_len][head_dim], 2 or 1 bytes
// These hooks look into a model prepared by ANOTHER instance of the backend (the product library: its decode stream is not this
// library's g_stream), so the copy is fenced by whole-device synchronisations on both sides and itself runs on this library's stream.
void foreign_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
	init_hip();
	HIP_CHECK(hipDeviceSynchronize());
	dev_copy_sync(dst, src, bytes, kind);
	HIP_CHECK(hipDeviceSynchronize());
}

struct KvGeom {
	int n_kv_heads, head_dim, seq_len, kv_dim, ebytes;
	size_t layer_bytes;
};
KvGeom kv_geom(struct Transformer* t) {
	KvGeom g;
	g.n_kv_heads = t->config.n_kv_heads, g.head_dim = t->config.head_dim, g.seq_len = t->config.seq_len;
	g.kv_dim = g.n_kv_heads * g.head_dim;
	g.ebytes = t->state.kvbits / 8;
	g.layer_bytes = (size_t)g.kv_dim * g.seq_len * g.ebytes;
	CALM_REQUIRE(t->state.key_cache && (g.ebytes == 1 || g.ebytes == 2), "transformer not prepared by the hip backend");
	return g;
}

} // namespace

extern "C" void calm_hip_read_kv(struct Transformer* t, int layer, int which, uint16_t* host) {
	// back in the reference's [seq_len][kv_dim] order, as binary16 patterns: an fp8 cache's e5m2 bytes are widened
	// (byte << 8 is the binary16 of the same value, src/infer.c:28-35)
	const KvGeom g = kv_geom(t);
	CALM_REQUIRE(layer >= 0 && layer < t->config.n_layers, "calm_hip_read_kv: no such layer");
	std::vector<unsigned char> tmp(g.layer_bytes);
	foreign_copy(tmp.data(), (char*)(which ? t->state.value_cache : t->state.key_cache) + (size_t)layer * g.layer_bytes, g.layer_bytes, hipMemcpyDeviceToHost);
	for (int h = 0; h < g.n_kv_heads; ++h) {
		for (int p = 0; p < g.seq_len; ++p) {
			uint16_t* dst = host + (size_t)p * g.kv_dim + h * g.head_dim;
			const size_t src = ((size_t)h * g.seq_len + p) * g.head_dim;
			if (g.ebytes == 2) {
				memcpy(dst, tmp.data() + src * 2, g.head_dim * 2);
			} else {
				for (int i = 0; i < g.head_dim; ++i) {
					dst[i] = (uint16_t)((uint16_t)tmp[src + i] << 8);
				}
			}
		}
	}
}

extern "C" void calm_hip_write_kv(struct Transformer* t, int layer, int which, const uint16_t* host) {
	// the inverse: a [seq_len][kv_dim] array of binary16 patterns into the backend's cache (an fp8 cache keeps the top byte:
	// the caller passes values that are exact in e5m2) -- lets a test start deep inside a long context without decoding its
	// way there
	const KvGeom g = kv_geom(t);
	CALM_REQUIRE(layer >= 0 && layer < t->config.n_layers, "calm_hip_write_kv: no such layer");
	std::vector<unsigned char> tmp(g.layer_bytes);
	for (int h = 0; h < g.n_kv_heads; ++h) {
		for (int p = 0; p < g.seq_len; ++p) {
			const uint16_t* src = host + (size_t)p * g.kv_dim + h * g.head_dim;
			const size_t dst = ((size_t)h * g.seq_len + p) * g.head_dim;
			if (g.ebytes == 2) {
				memcpy(tmp.data() + dst * 2, src, g.head_dim * 2);
			} else {
				for (int i = 0; i < g.head_dim; ++i) {
					tmp[dst + i] = (unsigned char)(src[i] >> 8);
				}
			}
		}
	}
	foreign_copy((char*)(which ? t->state.value_cache : t->state.key_cache) + (size_t)layer * g.layer_bytes, tmp.data(), g.layer_bytes, hipMemcpyHostToDevice);
	// the transposed value cache, when the backend keeps one (prepare_ctx: head size 128, knob "attn_vt", a window beyond the unsplit
	// kernel's contexts), sits behind the [position][dim] one in the same allocation -- seen here from the allocation's size (this
	// library shares no state with the product library the model was prepared by): [layer][kv_head][block of positions][head_dim][position in block]
	size_t alloc = 0;
	void* base = nullptr;
	HIP_CHECK(hipMemGetAddressRange((hipDeviceptr_t*)&base, &alloc, (hipDeviceptr_t)t->state.value_cache));
	const size_t kv_bytes = g.layer_bytes * t->config.n_layers;
	if (which && alloc >= 2 * kv_bytes) {
		for (int hd = 0; hd < g.kv_dim; ++hd) {
			for (int p = 0; p < g.seq_len; ++p) {
				const uint16_t v = host[(size_t)p * g.kv_dim + hd];
				const size_t dst = attn_vt_offset(hd, p, g.head_dim, g.seq_len, g.ebytes);
				if (g.ebytes == 2) {
					memcpy(tmp.data() + dst * 2, &v, 2);
				} else {
					tmp[dst] = (unsigned char)(v >> 8);
				}
			}
		}
		foreign_copy((char*)t->state.value_cache + kv_bytes + (size_t)layer * g.layer_bytes, tmp.data(), g.layer_bytes, hipMemcpyHostToDevice);
	}
}

extern "C" void calm_hip_read_moe(struct Transformer* t, int layer, int* experts, float* weights) {
	// routing of the LAST decode step at `layer`: the n_experts_ac expert ids in rank order and their softmax weights, as
	// k_ffn_up left them (src/infer.c:277-305) -- from state.exp: [layer][CALM_MAX_EXPERTS] weights, then as many expert ids
	CALM_REQUIRE(layer >= 0 && layer < t->config.n_layers && t->config.n_experts > 0, "calm_hip_read_moe: no such layer / not a mixture-of-experts model");
	CALM_REQUIRE(t->state.exp, "calm_hip_read_moe: transformer not prepared by the hip backend on one device (a model split over CALM_HIP_DEVICES stages keeps its routing per stage)");
	const int n = t->config.n_experts_ac;
	const float* w = t->state.exp + (size_t)layer * CALM_MAX_EXPERTS;
	const int* e = (const int*)(t->state.exp + (size_t)t->config.n_layers * CALM_MAX_EXPERTS) + (size_t)layer * CALM_MAX_EXPERTS;
	foreign_copy(experts, e, n * sizeof(int), hipMemcpyDeviceToHost);
	foreign_copy(weights, w, n * sizeof(float), hipMemcpyDeviceToHost);
}

namespace {
template <bool NT>
__global__ __launch_bounds__(256) void k_membench(const u32x4* src, size_t n16, unsigned* sink) {
	unsigned acc = 0;
	size_t stride = (size_t)gridDim.x * 256 * 8;
	for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n16; i += stride) {
		u32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			size_t j = i + (size_t)u * 256;
			if (j < n16) {
				v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
			} else {
				v[u] = (u32x4){0u, 0u, 0u, 0u};
			}
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
} // namespace

extern "C" double calm_hip_membench(size_t bytes, int nt, int iters) {
	init_hip();
	size_t n16 = bytes / 16;
	u32x4* buf = (u32x4*)dev_alloc(n16 * 16);
	unsigned* sink = (unsigned*)dev_alloc(4);
	dev_fill(buf, 0x5a, n16 * 16);
	int blocks = g_ncu * 8;
	auto go = [&]() {
		if (nt) {
			hipLaunchKernelGGL(k_membench<true>, dim3(blocks), dim3(256), 0, g_stream, buf, n16, sink);
		} else {
			hipLaunchKernelGGL(k_membench<false>, dim3(blocks), dim3(256), 0, g_stream, buf, n16, sink);
		}
	};
	go();
	hipEvent_t e0, e1;
	HIP_CHECK(hipEventCreate(&e0));
	HIP_CHECK(hipEventCreate(&e1));
	HIP_CHECK(hipEventRecord(e0, g_stream));
	for (int i = 0; i < iters; ++i) {
		go();
	}
	HIP_CHECK(hipEventRecord(e1, g_stream));
	HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	HIP_CHECK(hipEventDestroy(e0));
	HIP_CHECK(hipEventDestroy(e1));
	free_hip(buf), free_hip(sink);
	return (double)n16 * 16 * iters / 1e9 / ((double)ms / 1e3);
}
