// infer_hip.hip -- host side of the MI355X infer backend: the C ABI of include/calm_hip.h.
//
// Mirrors the role of the reference's src/infer.cu host code (upload_cuda :69-71, prepare_cuda
// :73-131, forward<T,KVT,AT> :651-741, perf_cuda :761-801) but not its structure: instead of one
// cooperative megakernel with 6-7 software grid barriers per layer (4-26 us each on this chip),
// a decode step is ~5 dependent weight-streaming kernels per layer, replayed from a hipGraph
// (kernel boundary ~1.2-1.9 us).  All per-token scalars travel through a device-resident
// TokState written by the graph's first kernel, whose arguments are patched before each replay.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/calm_hip.h"
#include "kernels.hip.h"
#include "prefill.hip.h"

using namespace calm;

#define HIP_CHECK(x)                                                                                                                              \
	do {                                                                                                                                          \
		hipError_t err_ = (x);                                                                                                                    \
		if (err_ != hipSuccess) {                                                                                                                 \
			fprintf(stderr, "HIP error in %s at %s:%d: %s (%s=%d)\n", __FUNCTION__, __FILE__, __LINE__, hipGetErrorString(err_), hipGetErrorName(err_), \
			        (int)err_);                                                                                                                   \
			abort();                                                                                                                              \
		}                                                                                                                                         \
	} while (0)

#define CALM_REQUIRE(cond, msg)                                                         \
	do {                                                                                \
		if (!(cond)) {                                                                  \
			fprintf(stderr, "calm_hip: %s (%s) at %s:%d\n", msg, #cond, __FILE__, __LINE__); \
			abort();                                                                    \
		}                                                                               \
	} while (0)

namespace {

constexpr int LDS_EXTRA = 2048 * (WG_WAVES > 4 ? 2 : 1); // reduction scratch + MoE routing scratch (k_ffn_up) / the waves' gate partials (k_attn_out) behind the activation image
constexpr int MAX_SPLIT = ATTN_MAX_SPLIT; // (k_attn_merge holds one partial per split in registers)
// the scratch behind the image: k_attn_out<GATE> parks every wave's ep + 2 partial sums there, k_ffn_up its reduction words, the router's
// logits and the picks (kernels.hip.h)
static_assert((size_t)WG_WAVES * (GATE_MAX_E + 2) * sizeof(float) <= LDS_EXTRA, "LDS_EXTRA: k_attn_out's gate partials of all waves");
static_assert((16 + (GATE_MAX_E + 2) + 2 * GATE_MAX_E) * sizeof(float) <= LDS_EXTRA, "LDS_EXTRA: k_ffn_up's routing scratch");

hipStream_t g_stream; // the CURRENT device's decode stream (multi-device: switched by use_dev)
int g_device = -1;
int g_ncu = 256;

// CALM_HIP_DEVICES=P (P > 1): the layers of every model prepared by this process are split over P pipeline stages, stage s on
// device (first + s) % visible devices -- one process, the four entry points unchanged, so the reference's CLI drives a sharded
// model as it drives one GPU (the reference itself pins device 0, src/infer.cu:79).  With fewer physical
// devices than stages several stages share a device (each on its own stream): the whole path runs, and is tested, on one GPU.
struct DevSlot {
	int dev = 0;
	hipStream_t stream = nullptr;
	int ncu = 256;
};
std::vector<DevSlot> g_devs; // one per stage; size 1 = the single-device backend
void use_dev(int s) {
	HIP_CHECK(hipSetDevice(g_devs[s].dev));
	g_stream = g_devs[s].stream;
	g_ncu = g_devs[s].ncu;
}
// multi-device: upload_hip cannot know which layer (hence which device) a tensor belongs to -- run.c uploads before it resolves
// names (src/run.c:550-561 then :563-576) -- so it only notes the host range and hands the host pointer back; prepare_hip, which
// sees struct Weights, uploads every tensor to its stage's device (the mapping stays valid until then: src/run.c:515,637).
std::map<const void*, size_t> g_pending_uploads;
// ... unless the host says where it belongs first: calm_hip_configure("stage", s) routes the following upload_hip / alloc_hip
// calls to stage s's device at once (-1: back to deferring) -- for hosts that know tensor names, or fill tensors on the device
int g_alloc_stage = -1;
double g_hop_us = 0; // time inside the stage-to-stage copies of profiled steps, and their number (perf_hip; calm_hip_query "handoff_ns" / "handoffs")
uint64_t g_hop_n = 0;
int g_use_graph = 1; // CALM_HIP_GRAPH=0 -> eager launches
int g_prof = 0;      // CALM_HIP_PROF=1 -> eager + per-stage events
int g_split_t = 128;   // kv positions per attention split up to 32 splits (two rounds of the 4-wave GQA kernel); twice that beyond
int g_split_min = 384; // contexts up to this many positions use the unsplit one-workgroup-per-head kernel
int g_attn_vt = 1;     // split attention (contexts beyond split_min) on the matrix cores over the transposed value cache where the head size is 128 (k_attn_vt); 0: k_attn_gqa
// "forms" (decode) and "pf_forms" (prompt ingestion): the launchers pick a kernel form per shape by rule; these two bit sets exist for the
// tests that run EVERY shipping form on small fixtures and compare (0 = the rules).  No form here is an experiment: each is what some
// BASELINE or public shape gets by rule.
//   forms     1: plain forms -- input vector read from LDS at every step (not held in registers: other widths), k_ffn_up's rounds dealt
//                evenly (no skew: other grids), one k_ffn_down pass per expert (not side by side: large experts)
//             2: small-matrix forms always -- k_qkv tiles half as deep, k_attn_out / k_ffn_down one row per task (by rule: TinyLlama, DBRX)
//             8: a mixture-of-experts layer's gate computed by every workgroup of k_ffn_up from the vector (MOE == 1; by rule: more
//                than 64 experts, a parallel residual) instead of derived from the partial sums k_attn_out's epilogue leaves (MOE == 2)
//   pf_forms  1: K-split GEMMs only (short prompts)          2: the big form for every dense FFN-up / classifier (long prompts)
//             4: grouped (expert) GEMMs never in the big form   8: ... always           (by rule: from 64 packed rows per expert)
//            16: lane-arithmetic prompt attention (head sizes other than 64 / 128)      32: no skinny chain for 3-4 token chunks
int g_forms = 0, g_pf_forms = 0;
constexpr int SKEW_PERCENT = 14; // k_ffn_up: more tasks for the first-dispatched workgroup of each CU than an even split gives it (skew_cut)
inline int knob_small() { return (g_forms & 2) ? 1 : 0; } // rows_balance_one / launch_qkv: 0 = by the rule, 1 = always
inline bool pf_wide_on() { return !(g_pf_forms & 1); }
inline int pf_big_mode() { return (g_pf_forms & 1) ? 0 : ((g_pf_forms & 2) ? 2 : 1); }        // 0 never, 1 by the rule, 2 always
inline int pf_moe_big_mode() { return (g_pf_forms & 4) ? 0 : ((g_pf_forms & 8) ? 2 : 1); }
long g_fused_steps = 0; // decode steps enqueued with the attention inside k_qkv's launch (calm_hip_query("fused_steps"): tests check the path they mean to test ran)
int g_qkv_attn = 1;    // short-context attention inside k_qkv's launch (kernels.hip.h k_qkv_attn) where the shape allows (fused_ok); 0: k_qkv, then k_attn
int g_pf_score_mb = 256; // MiB of logits scratch the scoring GEMM may use (prefill_logprobs_hip scores a chunk in blocks of that many rows; read when the scratch is allocated)
long g_pf_redone = 0;   // prompt tokens sent back through the serial path because an activation left the binary16 range
int g_pf_chunk_moe = PF_NT_MOE; // ... of a mixture-of-experts model (1024, 2048 or 4096 = PF_NT_MOE; read at the same moment)
int g_pf_chunk = PF_NT_DENSE; // tokens per prompt chunk of a dense model (read when a model's prompt buffers are allocated; PF_NT ... PF_NT_DENSE)
char g_devname[256] = "none";

// CALM_HIP_PROF_JSON=<path>: algorithmic bytes per kernel, accumulated over every decode step of the process and written at
// exit -- the role of the reference's PROF_TOKEN kernel argument (src/infer.cu:22,679,702,735), which tools/cudaprof.cu joins
// with the CUPTI kernel records into a BW column; here tools/prof_summary.py joins this file with the rocprofv3 kernel trace.
struct KernelBytes {
	uint64_t launches = 0, bytes = 0;
};
const char* g_prof_json = nullptr;
std::map<std::string, KernelBytes> g_kernel_bytes;

int env_int(const char* name, int dflt) {
	const char* v = getenv(name);
	return v && *v ? atoi(v) : dflt;
}

// Kernels issue unconditional, unclamped 16-byte loads that may run past the end of a row / vector
// (kernels.hip.h: stage_load, tile_load); every device buffer carries this much slack behind it.
constexpr size_t DEV_PAD = 64 * 1024;

// Host -> device through two pinned staging buffers: the host's copy into one overlaps the DMA out of the other, on the decode
// stream.  (A pageable hipMemcpy per tensor -- what upload_cuda does, src/infer.cu:69-71 -- moved a 7 GB model in ~25 s here.)
// Small tensors take the plain synchronous copy.  Ordered on g_stream; prepare_hip drains it before the host may unmap.
struct Stager {
	void* pin[2] = {nullptr, nullptr};
	hipEvent_t done[2];
	bool busy[2] = {false, false};
};
constexpr size_t STAGE_CHUNK = 32u << 20;
std::map<int, Stager> g_stagers; // per device

void dev_copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);

void staged_upload(void* dst, const void* src, size_t size) {
	if (size < (1u << 20)) {
		dev_copy_sync(dst, src, size, hipMemcpyHostToDevice);
		return;
	}
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	Stager& st = g_stagers[dev];
	if (!st.pin[0]) {
		for (int b = 0; b < 2; ++b) {
			HIP_CHECK(hipHostMalloc(&st.pin[b], STAGE_CHUNK, hipHostMallocDefault));
			HIP_CHECK(hipEventCreateWithFlags(&st.done[b], hipEventDisableTiming));
		}
	}
	int b = 0;
	for (size_t off = 0; off < size; off += STAGE_CHUNK, b ^= 1) {
		const size_t n = size - off < STAGE_CHUNK ? size - off : STAGE_CHUNK;
		if (st.busy[b]) {
			HIP_CHECK(hipEventSynchronize(st.done[b]));
		}
		memcpy(st.pin[b], (const char*)src + off, n);
		HIP_CHECK(hipMemcpyAsync((char*)dst + off, st.pin[b], n, hipMemcpyHostToDevice, g_stream));
		HIP_CHECK(hipEventRecord(st.done[b], g_stream));
		st.busy[b] = true;
	}
	// complete on return, like the hipMemcpy it replaces (src/infer.cu:69-71): a host that runs kernels of its own on another
	// stream over the uploaded tensor (tools/synth_fill_hip.hip's quantiser) must find it there; the copies WITHIN a tensor overlap
	HIP_CHECK(hipStreamSynchronize(g_stream));
}

void* dev_alloc(size_t size) {
	void* p = nullptr;
	HIP_CHECK(hipMalloc(&p, size + DEV_PAD));
	return p;
}

// ONE stream orders everything this library does on a device -- the reference's discipline (src/infer.cu:40,94).  g_stream is created
// hipStreamNonBlocking, so the NULL stream is NOT ordered against it, and a NULL-stream hipMemset of device memory returns before the
// fill has run: a fill "ahead of" a g_stream kernel could land after it (round 4's red GPU suite: a zeroed output of a finished
// x += W.v).  Hence no bare hipMemset / hipMemcpy anywhere under csrc/ (tests/test_abi.py lints for them): fills and copies go through
// these helpers, on g_stream.  The copies are complete on return (their host side may be a stack object or pageable memory).
void dev_fill(void* p, int byte, size_t bytes) {
	HIP_CHECK(hipMemsetAsync(p, byte, bytes, g_stream));
}
void dev_zero(void* p, size_t bytes) {
	dev_fill(p, 0, bytes);
}
void dev_copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
	HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream));
}

// compile-time dispatch on a run-time bool: f(std::true_type) / f(std::false_type)
template <class F>
inline void by_bool(bool b, F f) {
	if (b) {
		f(std::true_type());
	} else {
		f(std::false_type());
	}
}

// rows of n weights at DB bits are whole 1-KiB wave-loads (the kernels' FULL variant)
template <int DB>
inline bool rows_full(int n) {
	return (n / (128 / DB)) % 64 == 0;
}

// float4 registers per thread the staging prologue needs for an n-float vector at `block` threads
inline bool stage_v4(int n, int block) {
	return n <= 4 * 4 * block;
}

struct StageProf {
	double us = 0;
	uint64_t bytes = 0;
	uint64_t runs = 0;
};

struct GraphEntry {
	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
	hipGraphNode_t begin_node = nullptr;
};

struct Ctx {
	struct Transformer* t = nullptr;
	// shapes
	int dim = 0, hidden = 0, head_dim = 0, n_layers = 0, n_heads = 0, n_kv_heads = 0, vocab = 0, seq_len = 0;
	int q_dim = 0, kv_dim = 0, kv_mul = 0, n_experts = 0, n_active = 0, dbits = 0, kvbits = 0, lpr = 0;
	// device state
	float *x = nullptr, *xb = nullptr, *q = nullptr, *att = nullptr, *he = nullptr, *partial = nullptr, *logits_d = nullptr;
	SampleState* sample_st = nullptr;
	float *moe_w = nullptr, *rope_freq = nullptr;
	int *moe_e = nullptr, *next_tok = nullptr, *trace = nullptr, *trace_count = nullptr;
	float2 *rope_cs = nullptr, *rope_cs1 = nullptr;
	TokState* ts = nullptr;
	unsigned long long* gran = nullptr; // k_qkv_attn's hand-off granules: (q_dim + 2 kv_dim) x {value, tag}, one buffer for every layer
	unsigned* fuse_err = nullptr;       // device word its bounded waits raise (the head's output is NaN then; calm_hip_query("fuse_timeouts") reads it)
	unsigned fuse_salt = 0;             // perf_stage_hip only (FuseArgs::salt)
	void *kc = nullptr, *vc = nullptr;
	void* vt = nullptr; // the value cache once more, transposed: [layer][kv_head][head_dim][seq_len] (k_attn_vt); head size 128 only, behind vc in ONE allocation (prepare_ctx)
	size_t kv_layer_bytes = 0;
	int attn_chunk = 1 << 30; // cached positions per attention split of the step being enqueued (launch_attn_lpr)
	// The transposed value cache mirrors rows [0, vt_rows) of the value cache.  Only steps with split attention write it (k_qkv's
	// 2-byte scattered stores cost 1-3 % of a short-context token, profiles/r04_startup.txt); vt_sync brings it up to date first.
	int vt_rows = 0;
	bool write_vt = true; // the step being enqueued writes V^T (run_step; part of the graph key through n_split)
	// mixture-of-experts routing ahead of k_ffn_up (kernels.hip.h k_attn_out GATE): per layer a [dim][gate_ep] fp32 table
	// moegate[e][j] * ffn_norm[j] (+ gate_ep column sums), and the [gate_ep + 2][GATE_COLS] partial sums of the last k_attn_out
	float *gate_mt = nullptr, *gate_part = nullptr;
	int gate_ep = 0;
	float* logits_h = nullptr; // pinned host
	int trace_cap = 0;
	// batched prompt ingestion (allocated on first use): token-major [pf_nt][...] activations of one chunk
	int pf_nt = 0; // tokens per chunk, fixed when the buffers are allocated: the knob "pf_chunk_moe" for mixtures of experts, else "pf_chunk"
	float *pf_x = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_att = nullptr, *pf_h = nullptr, *pf_partial = nullptr;
	unsigned* pf_tile_count = nullptr;
	float2* pf_rope = nullptr;
	int* pf_tok = nullptr;
	// pinned host words the prompt kernels raise when an activation leaves the binary16 range (prefill.hip.h): one per chunk of a
	// prefill call (PF_FLAG_WORDS; chunks beyond share the last), so that the call redoes the prompt from the FIRST chunk that raised
	// one, not from its start; pf_flag_ptr[k] = word k as the device sees it (the persistent sources of the per-chunk symbol copies)
	unsigned* pf_flag = nullptr;
	std::vector<unsigned*> pf_flag_ptr;
	// ... of a mixture-of-experts model: gate logits, per-expert row lists, one expert's gathered rows
	float *pf_gate = nullptr, *pf_wsel = nullptr, *pf_xe = nullptr, *pf_y = nullptr;
	int *pf_rows = nullptr, *pf_colexp = nullptr, *pf_slot = nullptr;
	int pf_max_cols = 0;
	// ... when the caller wants the log-probability of every next token: logits of the chunk, targets, results
	int pf_score_nt = 0; // tokens per block of the scoring GEMM (pf_logits holds that many rows)
	float *pf_logits = nullptr, *pf_lp = nullptr;
	int* pf_target = nullptr;
	// graph cache: (n_split / two-round split form / fused attention, kv_only, sink, chained, argmax | sample | copy)
	std::map<std::tuple<int, int, int, int, int>, GraphEntry> graphs;
	// argument block of the begin-token kernel (patched per replay)
	struct BeginArgs {
		TokState* ts;
		int token;
		const int* tok_src;
		int pos, kv_sink, kv_pos, kv_len;
		float* x;
		const void* embed;
		int dim;
		const float* rope_freq;
		float2* rope_cs;
		int half_hd;
	} ba;
	void* ba_ptrs[13];
	// profiling
	StageProf prof[CALM_STAGE_COUNT];
	std::vector<hipEvent_t> events;
	double marker_us = 0; // what an event between two kernels costs the queue, from the last calibrated profiling step
};

std::map<struct Transformer*, Ctx*> g_ctx;
Ctx* g_prof_ctx = nullptr;

Ctx* ctx_of(struct Transformer* t) {
	auto it = g_ctx.find(t);
	CALM_REQUIRE(it != g_ctx.end(), "this transformer was not prepared with prepare_hip -- or it is split over CALM_HIP_DEVICES > 1 stages, which forward_hip drives and this entry point does not");
	return it->second;
}

// Choose a grid for `ntasks` wave-tasks at `wpb` waves per workgroup: everything resident if it fits,
// else a whole number b of workgroups per CU.  Two costs pull against each other -- idle wave-slots
// in the last round (rounds * waves / ntasks) and too few waves to keep HBM busy (a wave holds 16 KiB
// in flight; 4 waves per CU measured latency-bound: gf4 FFN-up ran at 2.6 TB/s on a "perfectly
// balanced" b = 1 grid) -- so the waste is weighted by (1 + 1/(2b)).
int pick_blocks(int ntasks, int wpb, int kernel_bpc = 0, int max_bpc = 0) {
	int bpc = kernel_bpc > 0 ? kernel_bpc : 2; // resident 256-thread workgroups per CU a grid is sized for: 2 measured better than 3 and 4, except where KShape names another
	if (max_bpc > 0 && bpc > max_bpc) {
		bpc = max_bpc;
	}
	int need = (ntasks + wpb - 1) / wpb;
	if (need <= g_ncu * bpc) {
		return need > 0 ? need : 1;
	}
	int best_b = 1;
	double best = 1e30;
	for (int b = 1; b <= bpc; ++b) {
		long waves = (long)g_ncu * b * wpb;
		long rounds = (ntasks + waves - 1) / waves;
		double cost = (double)(rounds * waves) / ntasks * (1.0 + 0.5 / b);
		if (cost <= best + 1e-9) {
			best = cost;
			best_b = b;
		}
	}
	return g_ncu * best_b;
}

// the matvec kernels with a dim-sized vector: WG_WAVES waves per workgroup (kernels.hip.h); 512-thread workgroups sit one per CU
int pick_blocks_wg(int ntasks, int kernel_bpc = 0) {
	return pick_blocks(ntasks, WG_WAVES, kernel_bpc, WG_THREADS >= 512 ? 1 : 0);
}

template <int DB>
size_t lds_bytes(int n) {
	return (size_t)xs_slots<DB>(n) * 16 + LDS_EXTRA;
}

// A kernel may use more than 48 KiB of dynamic LDS only once the function has been told so -- per device.  Every launcher does it at
// its launch site through launch_lds (round 6; a list of instantiations kept by hand in prepare_hip used to, and a form missing from
// it was a launch failure waiting for an unusual shape); what has been granted is remembered, so a launch costs one map lookup.
std::map<std::pair<int, const void*>, size_t> g_lds_granted;
template <class K>
void allow_lds(K kernel, size_t bytes) {
	if (bytes <= 48 * 1024) {
		return;
	}
	CALM_REQUIRE(bytes <= 160 * 1024, "activation vector does not fit the 160 KiB LDS");
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	size_t& have = g_lds_granted[{dev, reinterpret_cast<const void*>(kernel)}];
	if (have < bytes) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
		have = bytes;
	}
}
template <class K, class... Args>
void launch_lds(K kernel, dim3 grid, dim3 block, size_t lds, Args... args) {
	allow_lds(kernel, lds);
	hipLaunchKernelGGL(kernel, grid, block, lds, g_stream, args...);
}

// tasks the first-dispatched half of a two-workgroups-per-CU grid takes (kernels.hip.h task_range): whole ROUNDS of the half grid
// (a cut inside a round measured no gain where a whole-round cut gave 0.7 us: profiles/r04_startup.txt), (50 + skew / 2) % of them;
// for k_ffn_up where the kernel is 12-32 rounds long -- Mistral-7B's 14 (20.8 -> 20.4 us, 20.3 -> 19.6 on another box; whole token
// + 0.8-1.5 %), Mixtral's 28 (- 0.3 us); DBRX's 42 rounds and the classifier measured slower with it, coarser rounds cannot be cut
// finely enough (Llama-3 gf4: 7 rounds, already 4 + 3).  0: no skew
inline int skew_cut(int ntasks, int grid, int waves_per_block) {
	if ((g_forms & 1) || grid != 2 * g_ncu) {
		return 0;
	}
	const int stride = (grid / 2) * waves_per_block, rounds = (ntasks + stride - 1) / stride;
	if (rounds < 12 || rounds > 32) {
		return 0;
	}
	const int r_old = (int)(rounds * (100 + SKEW_PERCENT) / 200.0 + 0.5);
	return r_old >= rounds ? 0 : r_old * stride;
}

// the register-resident form of a dim-sized matvec (kernels.hip.h XREG): rows of exactly xreg_chunks() KiB -- 4096 input columns at
// fp8 (4 chunks) or gf4 (2), 2048 at fp16 (4): the BASELINE models' dim
template <int DB>
inline bool use_xreg(int n) {
	return !(g_forms & 1) && n == xreg_chunks<DB, true>() * 64 * (128 / DB);
}

// ---------------------------------------------------------------- stage launchers ---------------

template <int DB>
void launch_begin(Ctx* c) {
	int n = c->dim > c->head_dim / 2 ? c->dim : c->head_dim / 2;
	hipLaunchKernelGGL((k_begin_token<DB>), dim3((n + 255) / 256), dim3(256), 0, g_stream, c->ba.ts, c->ba.token, c->ba.tok_src, c->ba.pos, c->ba.kv_sink,
	                   c->ba.kv_pos, c->ba.kv_len, c->ba.x, c->ba.embed, c->ba.dim, c->ba.rope_freq, c->ba.rope_cs, c->ba.half_hd);
}

template <int KVB>
void launch_rotate_sink(Ctx* c) {
	int total = c->n_kv_heads * CALM_KV_SINKS * (c->head_dim / 2);
	hipLaunchKernelGGL((k_rotate_sink<KVB>), dim3((total + 255) / 256, c->n_layers), dim3(256), 0, g_stream, c->kc, c->rope_cs1, c->n_kv_heads, c->head_dim,
	                   c->seq_len, CALM_KV_SINKS);
}

template <int DB, int KVB>
void launch_qkv(Ctx* c, int l) {
	struct Config* p = &c->t->config;
	struct Weights* w = &c->t->weights;
	QkvArgs a;
	a.x = c->x;
	a.norm_w = w->rms_att_weight[l];
	a.wq = w->wq[l], a.wk = w->wk[l], a.wv = w->wv[l];
	a.bqkv = w->bqkv[l];
	a.q = c->q;
	a.kc = (char*)c->kc + (size_t)l * c->kv_layer_bytes;
	a.vc = (char*)c->vc + (size_t)l * c->kv_layer_bytes;
	a.vt = (c->vt && c->write_vt) ? (char*)c->vt + (size_t)l * c->kv_layer_bytes : nullptr;
	a.xb_dump = p->norm_par ? c->xb : nullptr;
	a.ts = c->ts;
	a.rope_cs = c->rope_cs;
	a.dim = c->dim, a.q_dim = c->q_dim, a.kv_dim = c->kv_dim, a.head_dim = c->head_dim, a.seq_len = c->seq_len;
	a.eps = p->norm_eps, a.clip = p->qkv_clip, a.ln = p->norm_ln;
	int ntasks = (c->q_dim + 2 * c->kv_dim) / KShape<DB, KS_QKV>::NR;
	dim3 grid(pick_blocks_wg(ntasks, KShape<DB, KS_QKV>::BPC)), block(WG_THREADS);
	{
		// pick_blocks settles on ONE workgroup per CU for the 3072 row pairs of the BASELINE shapes (1.5 rounds of two per CU); measured
		// (profiles/r04_startup.txt): right for fp8 (7.61 us; two per CU 8.15, three 7.62), wrong for gf4, whose half-size matrix wants the
		// eight waves per CU: 6.86 -> 6.37 us
		const int wgs = DB == 4 ? 2 : 0;
		if (wgs > 0 && (ntasks + WG_WAVES - 1) / WG_WAVES > g_ncu * wgs) {
			grid = dim3(g_ncu * wgs);
		}
	}
	size_t lds = lds_bytes<DB>(c->dim);
	// Tiles half as deep (a) for a matrix so small that a wave's share of it (all 2 x 4 x ncu waves of a full grid) is less than one tile
	// -- the wave starts multiplying after half the bytes (TinyLlama) -- and (b) for rows the full depth does not divide but half of it
	// does (DBRX's 6-KiB rows: 2 + 2 + 2 instead of 4 + a half-empty 4; 12.7 -> 11.8 us)
	constexpr int U = KShape<DB, KS_QKV>::U;
	const size_t per_wave = (size_t)(c->q_dim + 2 * c->kv_dim) * c->dim * DB / 8 / ((size_t)g_ncu * 2 * WG_WAVES);
	const int chunks = (c->dim / (128 / DB) + 63) / 64;
	const bool half = knob_small() ? true : (per_wave < (size_t)KShape<DB, KS_QKV>::NR * U * 1024 || (U > 1 && chunks % U != 0 && chunks % (U / 2) == 0));
	by_bool(stage_v4(c->dim, WG_THREADS), [&](auto V4) {
		by_bool(rows_full<DB>(c->dim), [&](auto FULL) {
			by_bool(half, [&](auto HALF) {
				launch_lds(k_qkv<DB, KVB, decltype(V4)::value ? 4 : 8, decltype(FULL)::value, decltype(HALF)::value>, grid, block, lds, a.x, a.norm_w, a.dim, a.q_dim,
				                   a.kv_dim, a);
			});
		});
	});
}

// Short-context attention rides in k_qkv's launch (kernels.hip.h k_qkv_attn) when: the knob is on; the weights are fp8 or fp16; the step's attention is unsplit and its cached rows fit one workgroup's registers (256 at head size 128, 512 at 64); heads
// of 64 or 128 (8 / 16 lanes per row; LPR 4 is not instantiated, larger heads would need two loads per granule poll); the input vector
// takes the 4-registers-per-thread staging and whole-KiB rows (every BASELINE shape; the other forms stay with the two launches rather
// than doubling the instantiations); the heads leave at least half of the CUs to the row engine (FUSE_LDS: one workgroup per CU).
// gf4 weights stay with the two launches (no instantiation): their k_qkv wants eight waves per CU (launch_qkv), the fused launch's row engine
// has four on ncu - n_heads CUs, and the two cancel (Llama-3-8B gf4, 8 layers: 2408-2439 tok/s against 2428-2432; profiles/r06_qkv_attn.txt).
template <int DB>
bool fused_ok(const Ctx* c, int kv_len, int n_split) {
	return DB != 4 && g_qkv_attn != 0 && c->gran && n_split == 1 && (c->lpr == 8 || c->lpr == 16) && kv_len <= fuse_max_kv(c->lpr) && stage_v4(c->dim, WG_THREADS) &&
	       rows_full<DB>(c->dim) && c->n_layers <= 256 && c->head_dim <= 4 * 64 && 2 * c->n_heads <= g_ncu;
}
// Every workgroup of the fused launch asks for more than half a CU's LDS, so a CU holds ONE: the attention workgroups get CUs of their
// own.  Beside a row-engine workgroup they put their 128 wave-loads of K / V rows into the CU's queue ahead of its weight tiles, that
// CU's rows came in 2 us after everybody else's, and every head waits for some of them (Mistral-7B fp8, 32 layers: 623-625 tok/s with
// shared CUs, 634-635 with this; 606-608 for the two launches).
constexpr size_t FUSE_LDS = 82 * 1024;

template <int DB, int KVB>
void launch_qkv_attn(Ctx* c, int l) {
	struct Config* p = &c->t->config;
	struct Weights* w = &c->t->weights;
	QkvArgs a;
	a.x = c->x;
	a.norm_w = w->rms_att_weight[l];
	a.wq = w->wq[l], a.wk = w->wk[l], a.wv = w->wv[l];
	a.bqkv = w->bqkv[l];
	a.q = c->q;
	a.kc = (char*)c->kc + (size_t)l * c->kv_layer_bytes;
	a.vc = (char*)c->vc + (size_t)l * c->kv_layer_bytes;
	a.vt = nullptr; // an unsplit step leaves the transposed cache to vt_sync (run_step)
	a.xb_dump = p->norm_par ? c->xb : nullptr;
	a.ts = c->ts;
	a.rope_cs = c->rope_cs;
	a.dim = c->dim, a.q_dim = c->q_dim, a.kv_dim = c->kv_dim, a.head_dim = c->head_dim, a.seq_len = c->seq_len;
	a.eps = p->norm_eps, a.clip = p->qkv_clip, a.ln = p->norm_ln;
	FuseArgs f;
	f.gran = c->gran, f.out = c->att, f.err = c->fuse_err;
	f.n_heads = c->n_heads, f.kv_mul = c->kv_mul;
	f.layer = (unsigned)l, f.salt = c->fuse_salt;
	const int ntasks = (c->q_dim + 2 * c->kv_dim) / KShape<DB, KS_QKV>::NR;
	int rows_grid = g_ncu - c->n_heads; // one workgroup per CU (FUSE_LDS), the row engine on every CU the heads leave
	if ((ntasks + WG_WAVES - 1) / WG_WAVES < rows_grid) {
		rows_grid = (ntasks + WG_WAVES - 1) / WG_WAVES;
	}
	const dim3 grid(c->n_heads + rows_grid), block(WG_THREADS);
	const size_t lds = FUSE_LDS;
	static_assert(FUSE_LDS >= fuse_lds_bytes(8) && FUSE_LDS >= fuse_lds_bytes(16), "FUSE_LDS: the attention role's scratch");
	CALM_REQUIRE(lds_bytes<DB>(c->dim) <= FUSE_LDS, "k_qkv_attn: the activation image of a 4096-column vector fits FUSE_LDS");
	constexpr int U = KShape<DB, KS_QKV>::U;
	const size_t per_wave = (size_t)(c->q_dim + 2 * c->kv_dim) * c->dim * DB / 8 / ((size_t)g_ncu * 2 * WG_WAVES);
	const int chunks = (c->dim / (128 / DB) + 63) / 64;
	const bool half = knob_small() ? true : (per_wave < (size_t)KShape<DB, KS_QKV>::NR * U * 1024 || (U > 1 && chunks % U != 0 && chunks % (U / 2) == 0));
	if constexpr (DB != 4) {
		by_bool(half, [&](auto HALF) {
			by_bool(c->lpr == 16, [&](auto L16) {
				launch_lds(k_qkv_attn<DB, KVB, decltype(HALF)::value, decltype(L16)::value ? 16 : 8>, grid, block, lds, a.x, a.norm_w, a.dim, a.q_dim, a.kv_dim, c->n_heads, a, f);
			});
		});
	}
}

// split attention on the matrix cores over the transposed value cache (k_attn_vt): prepare_hip keeps one for head size 128 and windows of whole 64-position blocks
inline bool attn_uses_vt(const Ctx* c) {
	return g_attn_vt && c->vt;
}

void launch_attn_merge(Ctx* c, int n_split, int stride) {
	const dim3 mblock((c->head_dim + 63) / 64 * 64);
	if (n_split <= 16) {
		hipLaunchKernelGGL(k_attn_merge<16>, dim3(c->n_heads), mblock, 0, g_stream, c->partial, c->att, c->head_dim, n_split, stride);
	} else if (n_split <= 32) {
		hipLaunchKernelGGL(k_attn_merge<32>, dim3(c->n_heads), mblock, 0, g_stream, c->partial, c->att, c->head_dim, n_split, stride);
	} else {
		hipLaunchKernelGGL(k_attn_merge<64>, dim3(c->n_heads), mblock, 0, g_stream, c->partial, c->att, c->head_dim, n_split, stride);
	}
}

// query heads of one kv head that share a workgroup of k_attn_vt: all of them up to 8 (the MFMA tile has 16 columns), else the
// largest divisor of kv_mul up to 8
inline int attn_vt_heads(int kv_mul) {
	for (int q = kv_mul < 8 ? kv_mul : 8; q > 1; --q) {
		if (kv_mul % q == 0) {
			return q;
		}
	}
	return 1;
}

template <int KVB, int LPR>
void launch_attn_lpr(Ctx* c, int l, int n_split) {
	AttnArgs a;
	a.q = c->q;
	a.kc = (char*)c->kc + (size_t)l * c->kv_layer_bytes;
	a.vc = (char*)c->vc + (size_t)l * c->kv_layer_bytes;
	a.out = c->att;
	a.partial = c->partial;
	a.ts = c->ts;
	a.head_dim = c->head_dim, a.kv_mul = c->kv_mul, a.seq_len = c->seq_len, a.n_split = n_split;
	a.pf_kv0 = 0, a.pf_stride = 0, a.pf_nb = 0;
	if (n_split == 1) {
		// short context: one 16-wave workgroup per query head, everything in one round, no merge pass
		// (8 or 4 waves per workgroup measured 4.5 / 5.0 us against 3.65: profiles/HISTORY.md section 5c)
		hipLaunchKernelGGL((k_attn<KVB, LPR, 16>), dim3(c->n_heads), dim3(1024), 0, g_stream, a.ts, a.q, a.kc, a.vc, a.head_dim, a.kv_mul, a.seq_len, a);
		return;
	}
	// long context: K/V rows loaded once per kv head for several query heads, kv range split, then merged
	if (attn_uses_vt(c)) { // the matrix-core form over the transposed value cache: a wave per tile of keys, all qh query heads at once
		const int qh = attn_vt_heads(c->kv_mul);
		const dim3 grid(c->n_kv_heads * (c->kv_mul / qh) * n_split), block(256);
		const void* vt = (const char*)c->vt + (size_t)l * c->kv_layer_bytes;
		by_bool(qh > 4, [&](auto WIDE) {
			hipLaunchKernelGGL((k_attn_vt<KVB, decltype(WIDE)::value ? 8 : 4>), grid, block, 0, g_stream, a.ts, a.q, a.kc, vt, a.head_dim, a.kv_mul, a.seq_len, a.n_split, qh, a);
		});
		launch_attn_merge(c, n_split, ATTN_VT_PSTRIDE);
		return;
	}
	const int qh = c->kv_mul % 4 == 0 ? 4 : (c->kv_mul % 2 == 0 ? 2 : 1);
	dim3 grid(c->n_kv_heads * (c->kv_mul / qh) * n_split), block(ATTN_GQA_BLOCK);
	// a split of at most two rounds (4 waves x 64 / LPR positions x 4 tiles each) asks for all its rows at once; the split length
	// of THIS step (Ctx::attn_chunk, set by run_step from kv_len; part of the graph key through attn_two)
	const int step = (ATTN_GQA_BLOCK / 64) * (64 / LPR) * 4;
	const bool two = c->attn_chunk <= 2 * step;
	by_bool(two, [&](auto TWO) {
		constexpr bool T = decltype(TWO)::value;
		if (qh == 4) {
			hipLaunchKernelGGL((k_attn_gqa<KVB, LPR, 4, T>), grid, block, 0, g_stream, a.ts, a.q, a.kc, a.vc, a.head_dim, a.kv_mul, a.seq_len, a.n_split, a);
		} else if (qh == 2) {
			hipLaunchKernelGGL((k_attn_gqa<KVB, LPR, 2, T>), grid, block, 0, g_stream, a.ts, a.q, a.kc, a.vc, a.head_dim, a.kv_mul, a.seq_len, a.n_split, a);
		} else {
			hipLaunchKernelGGL((k_attn_gqa<KVB, LPR, 1, T>), grid, block, 0, g_stream, a.ts, a.q, a.kc, a.vc, a.head_dim, a.kv_mul, a.seq_len, a.n_split, a);
		}
	});
	launch_attn_merge(c, n_split, c->head_dim + 2);
}

template <int KVB>
void launch_attn(Ctx* c, int l, int n_split) {
	switch (c->lpr) {
	case 4:
		return launch_attn_lpr<KVB, 4>(c, l, n_split);
	case 8:
		return launch_attn_lpr<KVB, 8>(c, l, n_split);
	case 16:
		return launch_attn_lpr<KVB, 16>(c, l, n_split);
	case 32:
		return launch_attn_lpr<KVB, 32>(c, l, n_split);
	case 64:
		return launch_attn_lpr<KVB, 64>(c, l, n_split);
	default:
		CALM_REQUIRE(false, "unsupported head_dim");
	}
}

// Should a matrix of `rows` rows be walked ONE row per task instead of `nr` (2)?  Yes when the row groups leave the last round of a
// full grid (`waves` waves) markedly emptier than single rows would: DBRX's 6144 rows are 3072 pairs = 1.5 rounds of 2048 waves
// (a quarter of the wave-slots idle) but exactly 3 rounds of single rows; TinyLlama's 2048 rows are half a round of pairs.
inline bool rows_balance_one(int rows, int nr, int waves, int knob) {
	if (knob) {
		return knob == 1;
	}
	auto eff = [&](int n) {
		const int tasks = rows / n, rounds = (tasks + waves - 1) / waves;
		return (double)tasks / ((double)rounds * waves);
	};
	return nr > 1 && eff(1) > eff(nr) + 0.1;
}

// k_attn_out's grid and row grouping (k_ffn_up's MOE == 2 form folds one partial column per workgroup of it)
template <int DB>
int attn_out_grid(const Ctx* c, bool* one_out = nullptr) {
	const bool one = rows_balance_one(c->dim, KShape<DB, KS_ATTN_OUT>::NR, g_ncu * 2 * WG_WAVES, knob_small());
	if (one_out) {
		*one_out = one;
	}
	return pick_blocks_wg(c->dim / (one ? 1 : KShape<DB, KS_ATTN_OUT>::NR), KShape<DB, KS_ATTN_OUT>::BPC);
}

// the router's logits travel from k_attn_out's epilogue to k_ffn_up ("forms" 8 turns it off): a mixture-of-experts model whose FFN
// normalises the residual k_attn_out completes (not a parallel-residual one, which feeds the FFN the attention norm's output)
template <int DB>
bool moe_route_ahead(const Ctx* c) {
	return !(g_forms & 8) && c->gate_mt && attn_out_grid<DB>(c) <= GATE_COLS;
}

template <int DB>
void launch_attn_out(Ctx* c, int l) {
	bool one = false;
	dim3 grid(attn_out_grid<DB>(c, &one)), block(WG_THREADS);
	size_t lds = lds_bytes<DB>(c->q_dim);
	const void* wo = c->t->weights.wo[l];
	const bool gate = moe_route_ahead<DB>(c);
	const float* mt = gate ? c->gate_mt + (size_t)l * ((size_t)c->dim + 1) * c->gate_ep : nullptr;
	by_bool(stage_v4(c->q_dim, WG_THREADS), [&](auto V4) {
		by_bool(rows_full<DB>(c->q_dim), [&](auto FULL) {
			by_bool(one, [&](auto ONE) {
				by_bool(gate, [&](auto GATE) {
					if constexpr (decltype(V4)::value && decltype(FULL)::value) {
						if (use_xreg<DB>(c->q_dim)) {
							launch_lds(k_attn_out<DB, 4, true, decltype(ONE)::value, decltype(GATE)::value, true>, grid, block, lds, c->x, c->att, wo, c->dim, c->q_dim, mt,
							                   c->gate_part, c->gate_ep);
							return;
						}
					}
					launch_lds(k_attn_out<DB, decltype(V4)::value ? 4 : 8, decltype(FULL)::value, decltype(ONE)::value, decltype(GATE)::value>, grid, block, lds, c->x,
					                   c->att, wo, c->dim, c->q_dim, mt, c->gate_part, c->gate_ep);
				});
			});
		});
	});
}

template <int DB>
void launch_ffn_up(Ctx* c, int l) {
	struct Config* p = &c->t->config;
	struct Weights* w = &c->t->weights;
	FfnUpArgs a;
	a.x = p->norm_par ? c->xb : c->x;
	a.norm_w = p->norm_par ? nullptr : w->rms_ffn_weight[l];
	a.w1 = w->w1[l], a.w3 = w->w3[l], a.moegate = w->moegate[l];
	// routing weights / expert ids of THIS layer: one slice per layer, so the routing of the last step stays inspectable
	// (calm_hip_read_moe in the test library; the reference keeps only the last layer's in state.exp, src/infer.c:423-424)
	a.he = c->he, a.moe_w = c->moe_w + (size_t)l * CALM_MAX_EXPERTS, a.moe_e = c->moe_e + (size_t)l * CALM_MAX_EXPERTS;
	a.dim = c->dim, a.hidden = c->hidden, a.n_experts = c->n_experts, a.n_active = c->n_active;
	a.eps = p->norm_eps, a.ln = p->norm_ln, a.gelu = p->act_gelu;
	int nact = c->n_active > 0 ? c->n_active : 1;
	int ntasks = nact * (c->hidden / (KShape<DB, KS_FFN_UP>::NR / 2));
	dim3 grid(pick_blocks_wg(ntasks, KShape<DB, KS_FFN_UP>::BPC)), block(WG_THREADS);
	size_t lds = lds_bytes<DB>(c->dim);
	// 0: dense; 1: the gate inside the kernel; 2: from k_attn_out's partials (its grid rides in the upper bits of n_experts)
	const int moe = c->n_experts > 0 ? (moe_route_ahead<DB>(c) ? 2 : 1) : 0;
	a.gate_c = nullptr;
	a.cut = moe == 1 ? 0 : skew_cut(ntasks, (int)grid.x, WG_WAVES);
	if (moe == 2) {
		a.moegate = c->gate_part;
		a.n_experts = c->n_experts | (attn_out_grid<DB>(c) << 8) | (c->gate_ep << 24);
		a.gate_c = c->gate_mt + (size_t)l * ((size_t)c->dim + 1) * c->gate_ep + (size_t)c->dim * c->gate_ep;
	}
	auto go = [&](auto kern) { launch_lds(kern, grid, block, lds, a.x, a.norm_w, a.w1, a.w3, a.moegate, a.dim, a.hidden, a.n_experts, a.n_active, a); };
	by_bool(stage_v4(c->dim, WG_THREADS), [&](auto V4) {
		by_bool(rows_full<DB>(c->dim), [&](auto FULL) {
			constexpr int V = decltype(V4)::value ? 4 : 8;
			constexpr bool F = decltype(FULL)::value;
			if constexpr (V == 4 && F) {
				if (use_xreg<DB>(c->dim) && moe != 1) {
					if (moe == 2) {
						go(k_ffn_up<DB, 4, true, 2, true>);
					} else {
						go(k_ffn_up<DB, 4, true, 0, true>);
					}
					return;
				}
			}
			if (moe == 2) {
				go(k_ffn_up<DB, V, F, 2>);
			} else if (moe == 1) {
				go(k_ffn_up<DB, V, F, 1>);
			} else {
				go(k_ffn_up<DB, V, F, 0>);
			}
		});
	});
}

// columns of w2 one k_ffn_down launch covers: all of hidden_dim while its fp32 image (plus gf4's word sums) fits the LDS,
// else the smallest number of equal whole-KiB column ranges that do (Qwen1.5-72B's 49152 -> 2 x 24576)
template <int DB>
int ffn_down_cols(int hidden) {
	const int unit = 64 * (128 / DB); // columns of a 1-KiB row chunk: ranges start on chunk boundaries
	int passes = 1;
	while (lds_bytes<DB>((hidden + passes - 1) / passes + unit) > 160 * 1024) {
		++passes;
	}
	if (passes == 1) {
		return hidden;
	}
	const int per = (hidden + passes - 1) / passes;
	return (per + unit - 1) / unit * unit;
}

// k_ffn_down with all active experts' images side by side (SEG)?  Only where they are small -- the premise of that form: OLMoE's
// eight 4-KiB vectors, not Mixtral's two of 57 KiB (measured slower there, profiles/r04_moe.txt) -- and hidden_dim is one column range
template <int DB>
bool ffn_down_segs(const Ctx* c, int kn) {
	return !(g_forms & 1) && c->n_active > 1 && kn == c->hidden && (size_t)c->n_active * xs_slots<DB>(kn) * 16 <= 96 * 1024;
}

template <int DB>
void launch_ffn_down(Ctx* c, int l) {
	constexpr int BLOCK = 512;
	const void* w2 = c->t->weights.w2[l];
	const int cols = ffn_down_cols<DB>(c->hidden);
	for (int k0 = 0; k0 < c->hidden; k0 += cols) {
		const int kn = c->hidden - k0 < cols ? c->hidden - k0 : cols;
		// tile depth: the format's shape, or 2 rows x 2 chunks for fp8 / fp16 rows of 4 n + 2 chunks -- hidden 14336 at fp8 = 14 -- walked
		// in exact steps of 2 instead of 4 + 4 + 4 + a half-empty 4 (- 1 us, profiles/r03_startup_experiments.txt; gf4's 7-chunk rows as one
		// 2 x 7 tile measured 12.8 us against 9.7 for the format's 2 x 2: removed in round 6)
		const int chunks = kn / (64 * (128 / DB));
		// ... or ONE row x 4 chunks when the matrix has fewer row pairs than a full grid has waves (half of them would get no task)
		const bool few_rows = rows_balance_one(c->dim, 2, g_ncu * (BLOCK / 64), knob_small());
		const int uo = few_rows ? 1 : ((DB != 4 && rows_full<DB>(kn) && chunks % 4 == 2) ? 2 : 0);
		int ntasks = c->dim / (uo == 1 ? 1 : (uo ? 2 : KShape<DB, KS_FFN_DOWN>::NR));
		dim3 grid(pick_blocks(ntasks, BLOCK / 64)), block(BLOCK);
		const bool segs = ffn_down_segs<DB>(c, kn);
		size_t lds = segs ? (size_t)c->n_active * xs_slots<DB>(kn) * 16 + LDS_EXTRA : lds_bytes<DB>(kn);
		auto go = [&](auto kern) {
			launch_lds(kern, grid, block, lds, c->x, c->he, w2, c->moe_w + (size_t)l * CALM_MAX_EXPERTS, c->moe_e + (size_t)l * CALM_MAX_EXPERTS, c->dim,
			                   c->hidden, c->n_active, k0, kn);
		};
		if (segs) {
			// small images: the 4-register staging form.  Steps of the largest of 4 / 2 / 1 chunks that divides a segment's chunk
			// count (a segment is walked in whole steps: OLMoE's 1-KiB rows in steps of 4 asked for 4 KiB per KiB used), one row
			// per task by the same rule as the dense form
			const bool full = rows_full<DB>(kn), one = few_rows;
			const int cpi = (kn / (128 / DB) + 63) / 64, u = cpi % 4 == 0 ? 4 : (cpi % 2 == 0 ? 2 : 1);
			ntasks = c->dim / (one ? 1 : 2);
			grid = dim3(pick_blocks(ntasks, BLOCK / 64));
			auto pick = [&](auto FULLc, auto ONEc) {
				constexpr bool F = decltype(FULLc)::value;
				constexpr int O = decltype(ONEc)::value ? 8 : 0;
				if (u == 4) {
					go(k_ffn_down<DB, BLOCK, 4, 4 + O, F, true>);
				} else if (u == 2) {
					go(k_ffn_down<DB, BLOCK, 4, 2 + O, F, true>);
				} else {
					go(k_ffn_down<DB, BLOCK, 4, 1 + O, F, true>);
				}
			};
			by_bool(full, [&](auto FULLc) { by_bool(one, [&](auto ONEc) { pick(FULLc, ONEc); }); });
			continue;
		}
		by_bool(stage_v4(kn, BLOCK), [&](auto V4) {
			constexpr int V = decltype(V4)::value ? 4 : 8;
			if (uo == 1 && rows_full<DB>(kn)) {
				go(k_ffn_down<DB, BLOCK, V, 1, true, false>);
			} else if (uo == 1) {
				go(k_ffn_down<DB, BLOCK, V, 1, false, false>);
			} else if (uo == 2) {
				go(k_ffn_down<DB, BLOCK, V, 2, true, false>);
			} else if (rows_full<DB>(kn)) {
				go(k_ffn_down<DB, BLOCK, V, 0, true, false>);
			} else {
				go(k_ffn_down<DB, BLOCK, V, 0, false, false>);
			}
		});
	}
}

template <int DB>
void launch_output(Ctx* c) {
	struct Config* p = &c->t->config;
	int ntasks = (c->vocab + KShape<DB, KS_OUTPUT>::NR - 1) / KShape<DB, KS_OUTPUT>::NR;
	dim3 grid(pick_blocks_wg(ntasks, KShape<DB, KS_OUTPUT>::BPC)), block(WG_THREADS);
	size_t lds = lds_bytes<DB>(c->dim);
	by_bool(stage_v4(c->dim, WG_THREADS), [&](auto V4) {
		by_bool(rows_full<DB>(c->dim), [&](auto FULL) {
			if constexpr (decltype(V4)::value && decltype(FULL)::value) {
				// (not the gf4 classifier: its 4 workgroups per CU no longer fit with 145 VGPRs -- 3 fit, the grid's last quarter ran as a
				// second batch: 46.9 against 45.0 us without, profiles/r04_gf4.txt)
				if (use_xreg<DB>(c->dim) && DB != 4) {
					launch_lds(k_output<DB, 4, true, true>, grid, block, lds, c->logits_d, c->x, c->t->weights.rms_final_weight, c->t->weights.wcls, c->dim, c->vocab,
					                   p->norm_eps, (int)p->norm_ln, 0);
					return;
				}
			}
			launch_lds(k_output<DB, decltype(V4)::value ? 4 : 8, decltype(FULL)::value>, grid, block, lds, c->logits_d, c->x, c->t->weights.rms_final_weight,
			                   c->t->weights.wcls, c->dim, c->vocab, p->norm_eps, (int)p->norm_ln, 0);
		});
	});
}

void launch_argmax(Ctx* c) {
	hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, g_stream, c->logits_d, c->vocab, c->next_tok, c->trace, c->trace_count);
}

void launch_sample(Ctx* c) {
	hipLaunchKernelGGL(k_sample_minp, dim3(1), dim3(1024), 0, g_stream, c->logits_d, c->vocab, c->next_tok, c->trace, c->trace_count, c->sample_st);
}

int attn_splits(const Ctx* c, int kv_len) {
	if (kv_len <= g_split_min) {
		return 1;
	}
	// 32 splits x 8 kv heads cover the chip once; past that, longer splits (more rounds per workgroup, loaded one
	// ahead) measured better than more workgroups: 8k context, 32 x 256 positions 12.8 us vs 64 x 128 14.4-16.6 us.
	// k_attn_vt over an e5m2 cache walks 4 waves x 64 positions per round: splits twice as long.
	const int t = (attn_uses_vt(c) && c->kvbits == 8) ? 2 * g_split_t : g_split_t;
	int n = (kv_len + t - 1) / t;
	if (n > 32) {
		int n2 = (kv_len + 2 * t - 1) / (2 * t);
		n = n2 > 32 ? n2 : 32;
	}
	return n > MAX_SPLIT ? MAX_SPLIT : n;
}

bool fused_step(const Ctx* c, int kv_len, int n_split) {
	return c->dbits == 16 ? fused_ok<16>(c, kv_len, n_split) : (c->dbits == 8 ? fused_ok<8>(c, kv_len, n_split) : fused_ok<4>(c, kv_len, n_split));
}

// algorithmic bytes per launch, the reference's accounting (src/infer.cu:685-699)
uint64_t stage_bytes(Ctx* c, int stage, int kv_len) {
	uint64_t db = c->dbits;
	uint64_t kvbw = (uint64_t)c->kv_dim * kv_len * (c->kvbits / 8);
	int nact = c->n_active > 0 ? c->n_active : 1;
	switch (stage) {
	case CALM_STAGE_QKV:
		return (uint64_t)(c->q_dim + 2 * c->kv_dim) * c->dim * db / 8;
	case CALM_STAGE_ATTN:
		return 2 * kvbw;
	case CALM_STAGE_ATTN_OUT:
		return (uint64_t)c->q_dim * c->dim * db / 8;
	case CALM_STAGE_FFN_UP:
		return 2 * ((uint64_t)c->hidden * c->dim * db / 8) * nact;
	case CALM_STAGE_FFN_DOWN:
		return ((uint64_t)c->hidden * c->dim * db / 8) * nact;
	case CALM_STAGE_OUTPUT:
		return (uint64_t)c->vocab * c->dim * db / 8;
	}
	return 0;
}

// rows [vt_rows, upto) of every layer's value cache -> the transposed cache (they were written by steps that skip it)
void vt_sync(Ctx* c, int upto) {
	if (!c->vt || c->vt_rows >= upto) {
		return;
	}
	const size_t layer_elems = (size_t)c->kv_dim * c->seq_len;
	// in blocks of rows whose element count stays far inside an int (the kernel indexes one thread per element: a 1M-position window
	// of 4096-wide rows would be 2^32 of them)
	const int max_rows = (1 << 30) / c->kv_dim;
	for (int r0 = c->vt_rows; r0 < upto; r0 += max_rows) {
		const int r1 = upto - r0 < max_rows ? upto : r0 + max_rows;
		const dim3 grid(((r1 - r0) * c->kv_dim + 255) / 256, c->n_layers);
		if (c->kvbits == 16) {
			hipLaunchKernelGGL(k_vt_backfill<16>, grid, dim3(256), 0, g_stream, c->vc, c->vt, layer_elems, c->kv_dim, c->head_dim, c->seq_len, r0, r1);
		} else {
			hipLaunchKernelGGL(k_vt_backfill<8>, grid, dim3(256), 0, g_stream, c->vc, c->vt, layer_elems, c->kv_dim, c->head_dim, c->seq_len, r0, r1);
		}
	}
	c->vt_rows = upto;
}

// ---------------------------------------------------------------- one decode step ---------------

struct StepPlan {
	int n_split;
	bool kv_only, sink, chained, argmax, copy_logits;
	bool sample; // min-p draw on the device (decode_sample_hip) instead of the arg-max
	bool fused;  // the attention inside k_qkv's launch (fused_ok; set by run_step)
};

struct Ctx;
void run_step(Ctx* c, int token, const int* tok_src, int pos, StepPlan sp, bool embed = true);

template <int DB, int KVB>
void enqueue_step(Ctx* c, const StepPlan& sp, bool timed) {
	size_t ev = 0;
	auto mark = [&]() {
		if (timed) {
			if (ev >= c->events.size()) {
				hipEvent_t e;
				HIP_CHECK(hipEventCreate(&e));
				c->events.push_back(e);
			}
			HIP_CHECK(hipEventRecord(c->events[ev++], g_stream));
		}
	};
	launch_begin<DB>(c);
	if (sp.sink) {
		launch_rotate_sink<KVB>(c);
	}
	for (int l = 0; l < c->n_layers; ++l) {
		mark();
		if (sp.fused) { // one launch, one span: a profiled step files it (and both stages' bytes) under the QKV stage
			launch_qkv_attn<DB, KVB>(c, l);
		} else {
			launch_qkv<DB, KVB>(c, l);
			mark();
			launch_attn<KVB>(c, l, sp.n_split);
		}
		mark();
		launch_attn_out<DB>(c, l);
		mark();
		launch_ffn_up<DB>(c, l);
		mark();
		launch_ffn_down<DB>(c, l);
	}
	mark();
	if (!sp.kv_only) {
		launch_output<DB>(c);
		mark();
		if (sp.sample) {
			launch_sample(c);
		} else if (sp.argmax) {
			launch_argmax(c);
		}
		if (sp.copy_logits) {
			HIP_CHECK(hipMemcpyAsync(c->logits_h, c->logits_d, (size_t)c->vocab * sizeof(float), hipMemcpyDeviceToHost, g_stream));
		}
	}
	HIP_CHECK(hipGetLastError());
}

void dispatch_step(Ctx* c, const StepPlan& sp, bool timed) {
#define CASE(db, kvb)                       \
	if (c->dbits == db && c->kvbits == kvb) \
	return enqueue_step<db, kvb>(c, sp, timed)
	CASE(16, 16);
	CASE(8, 16);
	CASE(4, 16);
	CASE(16, 8);
	CASE(8, 8);
	CASE(4, 8);
#undef CASE
	CALM_REQUIRE(false, "unsupported dbits/kvbits combination: dbits must be 4, 8 or 16, kvbits must be 8 or 16");
}

void account_step(Ctx* c, const StepPlan& sp, int kv_len) {
	auto add = [&](const char* kernel, uint64_t launches, uint64_t bytes_per_launch) {
		KernelBytes& k = g_kernel_bytes[kernel];
		k.launches += launches;
		k.bytes += launches * bytes_per_launch;
	};
	const uint64_t L = c->n_layers;
	if (sp.fused) { // one launch, both stages' bytes
		add("k_qkv_attn", L, stage_bytes(c, CALM_STAGE_QKV, kv_len) + stage_bytes(c, CALM_STAGE_ATTN, kv_len));
	} else {
		add("k_qkv", L, stage_bytes(c, CALM_STAGE_QKV, kv_len));
	}
	if (sp.fused) {
	} else if (sp.n_split == 1) {
		add("k_attn", L, stage_bytes(c, CALM_STAGE_ATTN, kv_len));
	} else {
		const bool vt = attn_uses_vt(c);
		add(vt ? "k_attn_vt" : "k_attn_gqa", L, stage_bytes(c, CALM_STAGE_ATTN, kv_len));
		add("k_attn_merge", L, (uint64_t)c->n_heads * sp.n_split * (vt ? ATTN_VT_PSTRIDE : c->head_dim + 2) * sizeof(float));
	}
	add("k_attn_out", L, stage_bytes(c, CALM_STAGE_ATTN_OUT, kv_len));
	add("k_ffn_up", L, stage_bytes(c, CALM_STAGE_FFN_UP, kv_len));
	add("k_ffn_down", L, stage_bytes(c, CALM_STAGE_FFN_DOWN, kv_len));
	if (!sp.kv_only) {
		add("k_output", 1, stage_bytes(c, CALM_STAGE_OUTPUT, kv_len));
	}
}

void write_prof_json() {
	if (!g_prof_json || g_kernel_bytes.empty()) {
		return;
	}
	FILE* f = fopen(g_prof_json, "w");
	if (!f) {
		return;
	}
	fprintf(f, "{");
	bool first = true;
	for (auto& kv : g_kernel_bytes) {
		fprintf(f, "%s\n \"%s\": {\"launches\": %llu, \"algorithmic_bytes\": %llu}", first ? "" : ",", kv.first.c_str(), (unsigned long long)kv.second.launches,
		        (unsigned long long)kv.second.bytes);
		first = false;
	}
	fprintf(f, "\n}\n");
	fclose(f);
}

void* begin_func(Ctx* c) {
	switch (c->dbits) {
	case 16:
		return (void*)k_begin_token<16>;
	case 8:
		return (void*)k_begin_token<8>;
	default:
		return (void*)k_begin_token<4>;
	}
}

void run_step(Ctx* c, int token, const int* tok_src, int pos, StepPlan sp, bool embed) {
	struct Config* p = &c->t->config;
	// rolling KV buffer with attention sinks (src/infer.c:329-332)
	int kv_sink = pos >= p->seq_len ? CALM_KV_SINKS : 0;
	int kv_pos = kv_sink + (pos - kv_sink) % (p->seq_len - kv_sink);
	int kv_len = pos >= p->seq_len ? p->seq_len : pos + 1;
	CALM_REQUIRE(tok_src || (token >= 0 && token < c->vocab), "token out of range");
	CALM_REQUIRE(!embed || c->t->weights.token_embedding_table, "this stage has no embedding table");
	CALM_REQUIRE(pos >= 0, "negative position");

	sp.sink = kv_sink > 0;
	sp.n_split = attn_splits(c, kv_len);
	sp.fused = fused_step(c, kv_len, sp.n_split);
	g_fused_steps += sp.fused;
	c->attn_chunk = (kv_len + sp.n_split - 1) / sp.n_split;
	const int attn_two = c->attn_chunk <= 2 * 4 * (64 / c->lpr) * 4; // the split kernel's two-round form (launch_attn_lpr): a different graph
	sp.chained = tok_src != nullptr;
	if (g_prof_json) {
		account_step(c, sp, kv_len);
	}
	// the transposed value cache: written by split steps only (their graphs; the unsplit kernel does not read it), after the rows
	// before this one have been brought up to date; an unsplit step leaves its row -- and so everything behind it -- unmirrored
	c->write_vt = c->vt && sp.n_split > 1;
	if (c->write_vt) {
		vt_sync(c, kv_sink ? c->seq_len : kv_pos);
		c->vt_rows = c->vt_rows > kv_pos + 1 ? c->vt_rows : kv_pos + 1;
	} else if (c->vt) {
		c->vt_rows = c->vt_rows < kv_pos ? c->vt_rows : kv_pos;
	}

	c->ba.token = token;
	c->ba.embed = embed ? c->t->weights.token_embedding_table : nullptr;
	c->ba.tok_src = tok_src;
	c->ba.pos = pos;
	c->ba.kv_sink = kv_sink;
	c->ba.kv_pos = kv_pos;
	c->ba.kv_len = kv_len;

	auto replay = [&]() { // the step from its hipGraph (captured on first use), begin-token arguments patched
		auto key = std::make_tuple((sp.n_split * 2 + attn_two) * 2 + (int)sp.fused, (int)sp.kv_only, (int)sp.sink, (int)sp.chained, (int)sp.sample * 4 + (int)sp.argmax * 2 + (int)sp.copy_logits);
		GraphEntry& ge = c->graphs[key];
		if (!ge.exec) {
			HIP_CHECK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
			dispatch_step(c, sp, false);
			HIP_CHECK(hipStreamEndCapture(g_stream, &ge.graph));
			HIP_CHECK(hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
			// the begin-token kernel is the only root of the (linear) graph
			size_t nroots = 0;
			HIP_CHECK(hipGraphGetRootNodes(ge.graph, nullptr, &nroots));
			CALM_REQUIRE(nroots == 1, "captured decode graph must have exactly one root");
			HIP_CHECK(hipGraphGetRootNodes(ge.graph, &ge.begin_node, &nroots));
			hipGraphNodeType ty;
			HIP_CHECK(hipGraphNodeGetType(ge.begin_node, &ty));
			CALM_REQUIRE(ty == hipGraphNodeTypeKernel, "root of the decode graph must be the begin-token kernel");
		}
		hipKernelNodeParams kp;
		memset(&kp, 0, sizeof(kp));
		int n = c->dim > c->head_dim / 2 ? c->dim : c->head_dim / 2;
		kp.func = begin_func(c);
		kp.gridDim = dim3((n + 255) / 256);
		kp.blockDim = dim3(256);
		kp.sharedMemBytes = 0;
		kp.kernelParams = c->ba_ptrs;
		kp.extra = nullptr;
		HIP_CHECK(hipGraphExecKernelNodeSetParams(ge.exec, ge.begin_node, &kp));
		HIP_CHECK(hipGraphLaunch(ge.exec, g_stream));
	};

	if (g_prof) {
		// Per-stage times from events between the kernels of an eager pass: per layer [qkv, attn, attn_out, ffn_up, ffn_down],
		// then [end-of-layers, after output].  Every event costs the queue a marker packet (2-5 us, partly hidden behind the
		// kernel before it), so the same step is first replayed from its graph between two events only: what the marked pass
		// takes beyond that, spread evenly over its spans, is taken off each of them.  (The step is idempotent: same token, same
		// position, same cache row written twice.)
		// Only a step WITHOUT side effects beyond its own cache row may run more than once: not past the rolling buffer (the sink
		// keys are re-rotated in place by every pass, src/infer.c:383-394), not a chained / sampled / arg-max step (next token,
		// trace and coin are consumed), not a later pipeline stage (x is consumed in place).  Any other step takes the one marked
		// pass and the marker cost of the last calibrated step -- profiling never changes results (src/infer.cu:761-801).
		const bool calibrate = !sp.sink && !sp.chained && !sp.sample && !sp.argmax && embed;
		float plain_ms = 0;
		if (calibrate) {
			while (c->events.size() < 2) {
				hipEvent_t e;
				HIP_CHECK(hipEventCreate(&e));
				c->events.push_back(e);
			}
			replay(); // (first use of this plan: captures the graph)
			HIP_CHECK(hipEventRecord(c->events[0], g_stream));
			replay();
			HIP_CHECK(hipEventRecord(c->events[1], g_stream));
			HIP_CHECK(hipStreamSynchronize(g_stream));
			HIP_CHECK(hipEventElapsedTime(&plain_ms, c->events[0], c->events[1]));
		}
		dispatch_step(c, sp, true);
		HIP_CHECK(hipStreamSynchronize(g_stream));
		const size_t nspans = (size_t)c->n_layers * (sp.fused ? 4 : 5) + (sp.kv_only ? 0 : 1);
		double marker_us = c->marker_us;
		if (calibrate) {
			float marked_ms = 0;
			HIP_CHECK(hipEventElapsedTime(&marked_ms, c->events[0], c->events[nspans]));
			marker_us = ((double)marked_ms - (double)plain_ms) * 1e3 / (double)nspans;
			marker_us = marker_us > 0 ? marker_us : 0;
			c->marker_us = marker_us;
		}
		size_t ev = 0;
		auto span = [&](int stage) {
			float ms = 0;
			HIP_CHECK(hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1]));
			ev++;
			const double us = ms * 1e3 - marker_us;
			c->prof[stage].us += us > 0.1 ? us : 0.1;
			c->prof[stage].bytes += stage_bytes(c, stage, kv_len);
			c->prof[stage].runs++;
		};
		for (int l = 0; l < c->n_layers; ++l) {
			span(CALM_STAGE_QKV);
			if (sp.fused) { // k_qkv_attn: the cached rows it read count under the launch that read them; no attention row of its own
				c->prof[CALM_STAGE_QKV].bytes += stage_bytes(c, CALM_STAGE_ATTN, kv_len);
			} else {
				span(CALM_STAGE_ATTN);
			}
			span(CALM_STAGE_ATTN_OUT), span(CALM_STAGE_FFN_UP), span(CALM_STAGE_FFN_DOWN);
		}
		if (!sp.kv_only) {
			span(CALM_STAGE_OUTPUT);
		}
		g_prof_ctx = c;
		return;
	}
	if (!g_use_graph) {
		dispatch_step(c, sp, false);
		return;
	}
	replay();
}

// ---------------------------------------------------------------- batched prompt ingestion -----

constexpr size_t PF_SPLIT_SLOTS = 2560; // 64-KiB partial tiles of a split launch (tiles x ranges: about one per CU, x 2-3 for grid padding; the big form's are four each)
constexpr int PF_SPLIT_TILES = 4096;
constexpr int PF_FLAG_WORDS = 256; // range-flag words per model: one per chunk of a prefill call (a 500k-token prompt in 2048-token chunks)

void pf_alloc(Ctx* c) {
	if (c->pf_x) {
		return;
	}
	// fragment-major matrices (prefill.hip.h: pf_unit) are sized in whole 64-column steps and zeroed once:
	// their padding is read as a multiplicand of zero weights and must stay finite
	auto frag = [&](int n, int rows) {
		size_t bytes = (size_t)rows * pf_steps(n) * 64 * sizeof(float);
		float* p = (float*)dev_alloc(bytes);
		dev_zero(p, bytes);
		return p;
	};
	// a mixture-of-experts chunk packs (token, expert) pairs into 64-row columns, one group per expert
	// (no chunk is longer than the context window -- a chunk never wraps the rolling buffer -- so a short window bounds the scratch: the
	// experts' gathered rows take chunk x active experts x (dim + hidden_dim) x 4 bytes, 1.7 GB for DBRX-132B at 4096 tokens)
	c->pf_nt = c->n_experts > 0 ? g_pf_chunk_moe : g_pf_chunk;
	c->pf_nt = std::min(c->pf_nt, std::max(PF_NT, (c->seq_len + 127) / 128 * 128));
	const int NT = c->pf_nt;
	// (worst case: every expert's group padded to a whole 128-row column pair -- k_pf_route gran 2 -- and the count even)
	c->pf_max_cols = c->n_experts > 0 ? ((NT * c->n_active + 63) / 64 + 2 * c->n_experts + 1) / 2 * 2 : 0;
	const int erows = c->n_experts > 0 ? c->pf_max_cols * 64 : NT;
	c->pf_x = (float*)dev_alloc((size_t)NT * c->dim * sizeof(float));
	c->pf_xn = frag(c->dim, NT);
	c->pf_q = (float*)dev_alloc((size_t)NT * c->q_dim * sizeof(float));
	c->pf_att = frag(c->q_dim, NT);
	c->pf_h = frag(c->hidden, erows);
	c->pf_rope = (float2*)dev_alloc((size_t)NT * (c->head_dim / 2) * sizeof(float2));
	c->pf_tok = (int*)dev_alloc(NT * sizeof(int));
	HIP_CHECK(hipHostMalloc((void**)&c->pf_flag, PF_FLAG_WORDS * sizeof(unsigned), hipHostMallocMapped));
	memset(c->pf_flag, 0, PF_FLAG_WORDS * sizeof(unsigned));
	unsigned* flag_dev = nullptr;
	HIP_CHECK(hipHostGetDevicePointer((void**)&flag_dev, c->pf_flag, 0));
	c->pf_flag_ptr.resize(PF_FLAG_WORDS);
	for (int k = 0; k < PF_FLAG_WORDS; ++k) {
		c->pf_flag_ptr[k] = flag_dev + k;
	}
	// k_pf_gemm_wide with K cut into ranges: partial tiles (64 KiB each) and the tiles' arrival counters (left at zero by every launch)
	c->pf_partial = (float*)dev_alloc(PF_SPLIT_SLOTS * 16384 * sizeof(float));
	c->pf_tile_count = (unsigned*)dev_alloc(PF_SPLIT_TILES * sizeof(unsigned));
	dev_zero(c->pf_tile_count, PF_SPLIT_TILES * sizeof(unsigned));
	if (c->n_experts > 0) {
		c->pf_gate = (float*)dev_alloc((size_t)NT * c->n_experts * sizeof(float));
		c->pf_rows = (int*)dev_alloc((size_t)erows * sizeof(int));
		c->pf_colexp = (int*)dev_alloc((size_t)c->pf_max_cols * sizeof(int));
		c->pf_slot = (int*)dev_alloc((size_t)NT * c->n_active * sizeof(int));
		c->pf_wsel = (float*)dev_alloc((size_t)NT * c->n_active * sizeof(float));
		c->pf_xe = frag(c->dim, erows);
		c->pf_y = (float*)dev_alloc((size_t)erows * c->dim * sizeof(float));
	}
}

template <int KVB, int LPR>
void launch_pf_attn_lpr(Ctx* c, int l, int nb, int pos0) {
	AttnArgs a;
	a.q = c->pf_q;
	a.kc = (char*)c->kc + (size_t)l * c->kv_layer_bytes;
	a.vc = (char*)c->vc + (size_t)l * c->kv_layer_bytes;
	a.out = c->pf_att;
	a.partial = nullptr;
	a.ts = c->ts;
	a.head_dim = c->head_dim, a.kv_mul = c->kv_mul, a.seq_len = c->seq_len, a.n_split = 1;
	a.pf_kv0 = pos0, a.pf_stride = c->q_dim, a.pf_nb = nb;
	constexpr int TQ = PfAttn<LPR>::TQ;
	hipLaunchKernelGGL((k_pf_attn<KVB, LPR>), dim3(c->n_heads, (nb + TQ - 1) / TQ), dim3(PF_ATTN_BLOCK), 0, g_stream, a);
}

// the matrix-core form (prefill.hip.h: k_pf_attn_mfma) for head sizes 64 / 128
template <int KVB, int HD>
void launch_pf_attn_mfma(Ctx* c, int l, int nb, int pos0) {
	AttnArgs a;
	a.q = c->pf_q;
	a.kc = (char*)c->kc + (size_t)l * c->kv_layer_bytes;
	a.vc = (char*)c->vc + (size_t)l * c->kv_layer_bytes;
	a.out = c->pf_att;
	a.partial = nullptr;
	a.ts = c->ts;
	a.head_dim = c->head_dim, a.kv_mul = c->kv_mul, a.seq_len = c->seq_len, a.n_split = 1;
	a.pf_kv0 = pos0, a.pf_stride = c->q_dim, a.pf_nb = nb;
	const int hg = c->kv_mul % 4 == 0 ? 4 : (c->kv_mul % 2 == 0 ? 2 : 1);
	const dim3 grid(c->n_kv_heads, (nb + 32 * (4 / hg) - 1) / (32 * (4 / hg)));
	auto go = [&](auto VT) { // VT: V read from the transposed cache (head size 128; a.vc is that cache)
		constexpr bool T = decltype(VT)::value;
		if (hg == 4) {
			hipLaunchKernelGGL((k_pf_attn_mfma<KVB, HD, 4, T>), grid, dim3(256), 0, g_stream, a);
		} else if (hg == 2) {
			hipLaunchKernelGGL((k_pf_attn_mfma<KVB, HD, 2, T>), grid, dim3(256), 0, g_stream, a);
		} else {
			hipLaunchKernelGGL((k_pf_attn_mfma<KVB, HD, 1, T>), grid, dim3(256), 0, g_stream, a);
		}
	};
	if constexpr (HD == 128) {
		if (c->vt && g_attn_vt) {
			a.vc = (char*)c->vt + (size_t)l * c->kv_layer_bytes;
			return go(std::true_type());
		}
	}
	go(std::false_type());
}

template <int KVB>
void launch_pf_attn(Ctx* c, int l, int nb, int pos0) {
	if (!(g_pf_forms & 16) && c->head_dim == 128) {
		return launch_pf_attn_mfma<KVB, 128>(c, l, nb, pos0);
	}
	if (!(g_pf_forms & 16) && c->head_dim == 64) {
		return launch_pf_attn_mfma<KVB, 64>(c, l, nb, pos0);
	}
	switch (c->lpr) {
	case 4:
		return launch_pf_attn_lpr<KVB, 4>(c, l, nb, pos0);
	case 8:
		return launch_pf_attn_lpr<KVB, 8>(c, l, nb, pos0);
	case 16:
		return launch_pf_attn_lpr<KVB, 16>(c, l, nb, pos0);
	case 32:
		return launch_pf_attn_lpr<KVB, 32>(c, l, nb, pos0);
	default:
		return launch_pf_attn_lpr<KVB, 64>(c, l, nb, pos0);
	}
}

// one chunk of nb <= Ctx::pf_nt tokens at positions pos0 .. pos0 + nb - 1 (no wrap of the rolling buffer):
// the layer loop of src/infer.c:349-458 with a token dimension
template <int DB, int KVB>
void prefill_chunk(Ctx* c, int nb, int pos0, bool score, bool embed) {
	struct Config* p = &c->t->config;
	struct Weights* w = &c->t->weights;
	const int half_hd = c->head_dim / 2;
	const int n0 = c->dim > half_hd ? c->dim : half_hd;
	hipLaunchKernelGGL((k_pf_begin<DB>), dim3((n0 + 255) / 256, nb), dim3(256), 0, g_stream, c->pf_tok, pos0, c->pf_x,
	                   embed ? w->token_embedding_table : nullptr, c->dim, c->rope_freq, c->pf_rope, half_hd);
	const dim3 block(256);
	const int cols = (nb + 63) / 64;
	// Two forms of the GEMM (prefill.hip.h): k_pf_gemm_wide (256 units x 64 tokens per workgroup, no K split inside the workgroup)
	// where its tiles cover enough of the chip, else the K-split form with S unit strips per wave (operands reused S times): the
	// S whose grid costs the least (rounds of workgroups over the CUs x a workgroup's time); ties go to the larger S.
	// Mixture of experts: the grouped GEMMs take the big form (128-token tiles: every expert's rows padded to whole 128-row columns by
	// k_pf_route, gran 2) where the chunk packs enough rows for the larger tile to pay for the extra padding (on average 64 rows per
	// expert): from an average of 64 packed rows per expert -- Mixtral-8x7B from 256 tokens (+ 6 %; + 17 % at 1536), DBRX-132B from 256
	// (+ 18 %), both behind at half that (profiles/r05_prefill.txt); fp8 / gf4 weights.  Otherwise 64-row columns and the wide / K-split forms.
	const int moe_gran = (c->n_experts > 0 && DB != 16 && pf_big_mode() && pf_moe_big_mode() && (pf_moe_big_mode() >= 2 || pf_big_mode() >= 2 || nb * c->n_active >= 64 * c->n_experts)) ? 2 : 1;
	// a grid of r rounds of workgroups over the CUs: whole rounds, and a last one that costs 0.6 of a round when it fills at most half
	// the slots (measured: profiles/r04_prefill.txt), a full one otherwise
	auto rounds_cost = [](double r) {
		const double f = r - floor(r);
		return floor(r) + (f < 1e-9 ? 0.0 : (f <= 0.5 ? 0.6 : 1.0));
	};
	auto gemm = [&](PfGemmArgs a, auto EPI, int ncols) {
		constexpr int epi = decltype(EPI)::value;
		constexpr int kvb = epi == PF_EPI_QKV ? KVB : 16; // only the QKV epilogue touches the cache
		// Chunks of 3 or 4 tokens (dense GEMMs): every weight streamed once at the decode kernels' rate with the tokens' activations behind
		// the stream (prefill.hip.h k_pf_skinny) -- where the image (4 x columns x 4 bytes) fits the LDS, a residual GEMM in column ranges.
		// Measured per layer on the 8-layer Mistral-7B fp8 shape (tools/smallchunk_bench.py; wall clock, eager launches included): 3 / 4
		// tokens 138 / 143 us against 184 / 180 through the GEMM forms; an eight-token variant of the same kernel lost (189-215 against 172).
		if constexpr (epi == PF_EPI_QKV || epi == PF_EPI_RESID || epi == PF_EPI_FFN_UP) {
			constexpr int CC = 16 * Fmt<DB>::G; // columns per chunk
			constexpr int T = 4;
			const int fit = (int)((150 * 1024) / (T * 4)) / CC * CC; // columns whose image fits
			if (!(g_pf_forms & 32) && nb <= T && !a.col_expert && a.K % CC == 0 && a.M % 4 == 0 && (epi == PF_EPI_RESID || a.K <= fit)) {
				const int ranges = (a.K + fit - 1) / fit;
				const int per = ((a.K / CC + ranges - 1) / ranges) * CC;
				for (int k0 = 0; k0 < a.K; k0 += per) {
					const int kn = a.K - k0 < per ? a.K - k0 : per;
					const size_t lds = (size_t)T * kn * 4;
					const int ngroups = a.M / 4;
					const int per_cu = lds <= 72 * 1024 ? 2 : 1;
					int blocks = (ngroups + 3) / 4;
					blocks = blocks > g_ncu * per_cu ? g_ncu * per_cu : blocks;
					auto kern = k_pf_skinny<DB, kvb, epi, T>;
					allow_lds(kern, lds);
					hipLaunchKernelGGL(kern, dim3(blocks), block, lds, g_stream, a, k0, kn);
				}
				return;
			}
		}
		// The big form (prefill.hip.h k_pf_gemm_big: 512 units x 128 tokens per 8-wave workgroup, one per CU) for the GEMMs with enough
		// units to fill the chip with such tiles -- the FFN-up and the classifier of a dense model from ~384 tokens: ahead of the wide
		// form from 5/8 of the CUs covered (+6...20 %), behind below that (tools/experiments/exp_pfgemm_big.hip, profiles/r04_prefill.txt).
		// fp8 / gf4 weights.  ("pf_forms" 1: never, 2: whatever the grid -- the tests' switches.)
		// The residual GEMM with long rows and few units (the FFN-down: 8 x 8 such tiles at 1024 tokens) takes it with K cut into 2-4 ranges,
		// one workgroup each, while a range keeps >= 48 steps (316 -> us at 1024 tokens).
		if constexpr ((epi == PF_EPI_FFN_UP || epi == PF_EPI_STORE || epi == PF_EPI_RESID) && DB != 16) {
			const int nxb = (a.M + PfBig<epi>::UNITS - 1) / PfBig<epi>::UNITS, ncb = (a.nb + PfBig<epi>::TOKENS - 1) / PfBig<epi>::TOKENS;
			int kr = 1;
			const bool grouped = a.col_expert != nullptr;
			if (grouped && moe_gran != 2) {
				kr = 0; // (64-row expert groups: the wide / K-split forms below)
			} else if (grouped && epi == PF_EPI_STORE) {
				// the experts' FFN-down (4096-6144 units: 8-12 tiles per token column): ranges of K by the rounds they leave, one workgroup
				// per CU; the columns that will be live: the packed rows plus half a column of padding per expert
				const int live = std::min(ncb, (nb * c->n_active + 127) / 128 + (c->n_experts + 1) / 2);
				const double r1 = (double)nxb * live / g_ncu;
				const size_t tiles_b = (size_t)8 * ((nxb + 7) / 8) * ncb;
				double best = rounds_cost(r1);
				for (int k = 2; k <= 4; k *= 2) {
					if (pf_steps(a.K) / k < 48 || tiles_b * k * 4 > PF_SPLIT_SLOTS || tiles_b > (size_t)PF_SPLIT_TILES) {
						break;
					}
					const double t = rounds_cost(r1 * k) / k * (1.0 + 0.03 * k);
					if (t < 0.97 * best) {
						best = t, kr = k;
					}
				}
			} else if (epi == PF_EPI_RESID) {
				kr = g_ncu / (nxb * ncb);
				kr = kr > 4 ? 4 : kr;
				kr = kr > pf_steps(a.K) / 48 ? pf_steps(a.K) / 48 : kr;
				const size_t tiles_b = (size_t)8 * ((nxb + 7) / 8) * ncb;
				if (kr < 2 || tiles_b * kr * 4 > PF_SPLIT_SLOTS || tiles_b > (size_t)PF_SPLIT_TILES) {
					kr = 0; // (the residual GEMM only in ranges: unsplit, its 4096 units are a quarter of the chip)
				}
			}
			if (pf_big_mode() && kr >= 1 && (pf_big_mode() >= 2 || grouped || (long)nxb * ncb * kr * 8 >= (long)g_ncu * 5)) {
				a.ncols = ncb, a.ksplit = kr, a.partial = c->pf_partial, a.tile_count = c->pf_tile_count;
				auto kern = k_pf_gemm_big<DB, epi>;
				allow_lds(kern, PfBigA<DB>::LDS_BYTES);
				hipLaunchKernelGGL(kern, dim3(pf_wide_grid(nxb, ncb, kr)), dim3(512), PfBigA<DB>::LDS_BYTES, g_stream, a);
				return;
			}
		}
		const int nx = (a.M + PfWide<epi>::UNITS - 1) / PfWide<epi>::UNITS;
		const int tiles = 8 * ((nx + 7) / 8) * ncols, nsteps = pf_steps(a.K);
		// Which form (profiles/r02_prefill_gemm.txt: Mistral-7B and TinyLlama shapes at 64-1024 tokens): the wide form from 5/8 of
		// the CUs busy; below that, the wide form with K cut into ranges (one workgroup each, the last to arrive folds the partial
		// tiles) where rows are long (the FFN-down: 85 against 119 us at 256 tokens), for the FFN-up, and x 2 from half the CUs;
		// otherwise the K-split form.
		int ks = 1; // 0: the K-split form
		if ((long)nx * ncols * 8 < (long)g_ncu * 5) {
			ks = 0;
			// (round 6: also from three token columns whatever the row length -- the QKV / wo GEMMs of 129-448 token prompts: Mistral-7B
			// 256 tokens 20.2 -> 21.5 k tok/s, 384 tokens 20.1 -> 21.1; at 64 tokens the K-split form stays ahead by 6 %, at 128 they tie)
			if (nsteps >= 80 || epi == PF_EPI_FFN_UP || nx * ncols * 2 >= g_ncu || ncols >= 3) {
				int k = g_ncu / (nx * ncols);
				k = k > 8 ? 8 : k;
				k = k > nsteps / 16 ? nsteps / 16 : k;
				if (k >= 2 && (size_t)tiles * k <= PF_SPLIT_SLOTS && tiles <= PF_SPLIT_TILES) {
					ks = k;
				}
			}
		} else {
			// Enough tiles -- but the workgroups run in rounds of two per CU, and what is left for the last round runs alone: Mixtral's
			// grouped FFN-down at 1024 tokens is 1.25 rounds (886 us, 271 TFLOP/s).  Ranges of K (one workgroup each, as above) make the
			// rounds finer: 2 or 4 where the model below says that saves more than the fold costs (~3 % per range).  A last round that fills
			// at most half the slots leaves one workgroup per CU, which runs at ~0.6 of a full round's time, not 0.5 (hence nothing for the
			// QKV GEMM's 1.5 rounds at 2048 tokens: measured 32.0 against 32.4 k tok/s with it split); a grid under one round gains nothing
			// from ranges at all (TinyLlama's FFN-down: - 5 %; round 6: the FFN-up of a 128-token prompt, 0.44 rounds, in two ranges: 88.7 -> 89.7 us --
			// a workgroup's step costs the CU the same whoever shares it).  profiles/r04_prefill.txt, r06_prefill.txt.
			const double r1 = (double)nx * ncols / (2.0 * g_ncu);
			double best = rounds_cost(r1);
			for (int k = 2; k <= 4 && r1 > 1.0; k *= 2) {
				if (nsteps / k < 16 || (size_t)tiles * k > PF_SPLIT_SLOTS || tiles > PF_SPLIT_TILES) {
					break;
				}
				const double t = rounds_cost(r1 * k) / k * (1.0 + 0.03 * k);
				if (t < 0.97 * best) {
					best = t, ks = k;
				}
			}
		}
		if (pf_wide_on() && ks >= 1) {
			a.ncols = ncols;
			a.ksplit = ks, a.partial = c->pf_partial, a.tile_count = c->pf_tile_count;
			auto kern = k_pf_gemm_wide<DB, kvb, epi, 1>;
			allow_lds(kern, PfWideA<DB>::LDS_BYTES); // (an attribute of the function on the current device)
			hipLaunchKernelGGL(kern, dim3(pf_wide_grid(nx, ncols, ks)), block, PfWideA<DB>::LDS_BYTES, g_stream, a);
			return;
		}
		if constexpr (epi == PF_EPI_FFN_UP) {
			hipLaunchKernelGGL((k_pf_gemm<DB, kvb, epi, 1>), dim3((a.M + 31) / 32, ncols), block, 0, g_stream, a);
		} else {
			int best = 1;
			long best_cost = 0;
			for (int S = 1; S <= 3; ++S) { // a workgroup's time grows like 2 + S: the B operand is fetched whatever S is (measured 27 : 40 : 46)
				long wgs = (long)((a.M + 32 * S - 1) / (32 * S)) * ncols;
				long cost = (wgs + g_ncu - 1) / g_ncu * (2 + S);
				if (S == 1 || cost <= best_cost) {
					best = S, best_cost = cost;
				}
			}
			const dim3 grid((a.M + 32 * best - 1) / (32 * best), ncols);
			if (best == 3) {
				hipLaunchKernelGGL((k_pf_gemm<DB, kvb, epi, 3>), grid, block, 0, g_stream, a);
			} else if (best == 2) {
				hipLaunchKernelGGL((k_pf_gemm<DB, kvb, epi, 2>), grid, block, 0, g_stream, a);
			} else {
				hipLaunchKernelGGL((k_pf_gemm<DB, kvb, epi, 1>), grid, block, 0, g_stream, a);
			}
		}
	};
	using EpiQkv = std::integral_constant<int, PF_EPI_QKV>;
	using EpiResid = std::integral_constant<int, PF_EPI_RESID>;
	using EpiUp = std::integral_constant<int, PF_EPI_FFN_UP>;
	PfGemmArgs a;
	memset(&a, 0, sizeof(a));
	a.nb = nb;
	a.rope = c->pf_rope;
	a.q_dim = c->q_dim, a.kv_dim = c->kv_dim, a.head_dim = c->head_dim, a.seq_len = c->seq_len, a.kv_pos0 = pos0;
	a.clip = p->qkv_clip, a.gelu = p->act_gelu;
	for (int l = 0; l < c->n_layers; ++l) {
		// attention norm; q / k / v + bias + clip + RoPE + KV append   (src/infer.c:352-381)
		hipLaunchKernelGGL(k_pf_norm, pf_norm_grid(nb, c->dim, g_ncu), block, 0, g_stream, (float4*)c->pf_xn, c->pf_x, w->rms_att_weight[l], c->dim, p->norm_eps, (int)p->norm_ln, nb);
		a.xin = (const float4*)c->pf_xn, a.K = c->dim, a.M = c->q_dim + 2 * c->kv_dim;
		a.w0 = w->wq[l], a.w1 = w->wk[l], a.w2 = w->wv[l], a.bqkv = w->bqkv[l];
		a.out = c->pf_q;
		a.kc = (char*)c->kc + (size_t)l * c->kv_layer_bytes, a.vc = (char*)c->vc + (size_t)l * c->kv_layer_bytes;
		a.vt = c->vt ? (char*)c->vt + (size_t)l * c->kv_layer_bytes : nullptr;
		gemm(a, EpiQkv(), cols);
		// causal attention of every token of the chunk over the cache (its own row included)
		launch_pf_attn<KVB>(c, l, nb, pos0);
		// x += wo . att   (src/infer.c:408-415)
		a.xin = (const float4*)c->pf_att, a.K = c->q_dim, a.M = c->dim, a.w0 = w->wo[l], a.out = c->pf_x;
		gemm(a, EpiResid(), cols);
		// FFN   (src/infer.c:417-457); parallel-residual models reuse the attention norm's output
		if (!p->norm_par) {
			hipLaunchKernelGGL(k_pf_norm, pf_norm_grid(nb, c->dim, g_ncu), block, 0, g_stream, (float4*)c->pf_xn, c->pf_x, w->rms_ffn_weight[l], c->dim, p->norm_eps, (int)p->norm_ln, nb);
		}
		if (c->n_experts == 0) {
			a.xin = (const float4*)c->pf_xn, a.K = c->dim, a.M = c->hidden, a.w0 = w->w1[l], a.w1 = w->w3[l], a.out = c->pf_h;
			gemm(a, EpiUp(), cols);
			a.xin = (const float4*)c->pf_h, a.K = c->hidden, a.M = c->dim, a.w0 = w->w2[l], a.out = c->pf_x;
			gemm(a, EpiResid(), cols);
			continue;
		}
		// mixture of experts (src/infer.c:422-457): gate logits of every token, routing into packed per-expert row
		// groups, ONE grouped GEMM per matrix over all experts' rows, then the experts' outputs are added to the
		// residual in rank order
		a.xin = (const float4*)c->pf_xn, a.K = c->dim, a.M = c->n_experts, a.w0 = w->moegate[l], a.out = c->pf_gate;
		hipLaunchKernelGGL((k_pf_gemm<DB, 16, PF_EPI_STORE, 1>), dim3((a.M + 31) / 32, cols), block, 0, g_stream, a);
		const int ecols = ((nb * c->n_active + 63) / 64 + moe_gran * c->n_experts + moe_gran - 1) / moe_gran * moe_gran; // worst case for this chunk
		hipLaunchKernelGGL(k_pf_route, dim3(1), dim3(PF_NT), 0, g_stream, c->pf_gate, nb, c->n_experts, c->n_active, ecols, moe_gran, c->pf_rows, c->pf_colexp,
		                   c->pf_slot, c->pf_wsel);
		hipLaunchKernelGGL(k_pf_gather, dim3(ecols * 2, (pf_steps(c->dim) + 7) / 8), block, 0, g_stream, (float4*)c->pf_xe, (const float4*)c->pf_xn, c->pf_rows, c->pf_colexp,
		                   c->dim);
		PfGemmArgs m = a;
		m.nb = ecols * 64;
		m.col_expert = c->pf_colexp, m.expert_stride = (size_t)c->hidden * c->dim * DB / 8;
		m.xin = (const float4*)c->pf_xe, m.K = c->dim, m.M = c->hidden, m.w0 = w->w1[l], m.w1 = w->w3[l], m.out = c->pf_h;
		gemm(m, EpiUp(), ecols);
		m.xin = (const float4*)c->pf_h, m.K = c->hidden, m.M = c->dim, m.w0 = w->w2[l], m.out = c->pf_y;
		gemm(m, std::integral_constant<int, PF_EPI_STORE>(), ecols);
		hipLaunchKernelGGL(k_pf_combine, dim3(nb), block, 0, g_stream, c->pf_x, c->pf_y, c->pf_slot, c->pf_wsel, c->n_active, c->dim);
	}
	if (score) {
		// final norm + classifier for every token of the chunk (src/infer.c:465-469), then log softmax of the target
		hipLaunchKernelGGL(k_pf_norm, pf_norm_grid(nb, c->dim, g_ncu), block, 0, g_stream, (float4*)c->pf_xn, c->pf_x, w->rms_final_weight, c->dim, p->norm_eps, (int)p->norm_ln, nb);
		// in blocks of pf_score_nt tokens (whole 128-token columns: a block of the fragment-major matrix starts at a 32-token group), so
		// that the logits scratch stays bounded for 128k / 256k vocabularies
		a.K = c->dim, a.M = c->vocab, a.w0 = w->wcls, a.out = c->pf_logits;
		for (int t0 = 0; t0 < nb; t0 += c->pf_score_nt) {
			const int n = nb - t0 < c->pf_score_nt ? nb - t0 : c->pf_score_nt;
			a.xin = (const float4*)c->pf_xn + (size_t)(t0 >> 5) * pf_steps(c->dim) * 512;
			a.nb = n;
			gemm(a, std::integral_constant<int, PF_EPI_STORE>(), (n + 63) / 64);
			hipLaunchKernelGGL(k_pf_logprob, dim3(n), block, 0, g_stream, c->pf_logits, c->vocab, c->pf_target + t0, c->pf_lp + t0);
		}
	}
	HIP_CHECK(hipGetLastError());
}

void dispatch_prefill_chunk(Ctx* c, int nb, int pos0, bool score, bool embed = true) {
#define CASE(db, kvb)                       \
	if (c->dbits == db && c->kvbits == kvb) \
	return prefill_chunk<db, kvb>(c, nb, pos0, score, embed)
	CASE(16, 16);
	CASE(8, 16);
	CASE(4, 16);
	CASE(16, 8);
	CASE(8, 8);
	CASE(4, 8);
#undef CASE
	CALM_REQUIRE(false, "unsupported dbits/kvbits combination");
}

} // namespace

// ================================================================ C ABI =======================

extern "C" int calm_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		return 0;
	}
	return n;
}

extern "C" const char* calm_hip_device_name(void) {
	return g_devname;
}

extern "C" int calm_hip_configure(const char* key, int value) {
	init_hip();
	HIP_CHECK(hipStreamSynchronize(g_stream));
	int* slot = nullptr;
	if (!strcmp(key, "graph")) {
		slot = &g_use_graph;
	} else if (!strcmp(key, "prof")) {
		slot = &g_prof;
	} else if (!strcmp(key, "split_t")) {
		slot = &g_split_t;
	} else if (!strcmp(key, "split_min")) {
		slot = &g_split_min;
	} else if (!strcmp(key, "attn_vt")) {
		slot = &g_attn_vt;
	} else if (!strcmp(key, "forms")) {
		slot = &g_forms;
	} else if (!strcmp(key, "pf_forms")) {
		slot = &g_pf_forms;
	} else if (!strcmp(key, "qkv_attn")) {
		slot = &g_qkv_attn;

	} else if (!strcmp(key, "pf_chunk")) {
		CALM_REQUIRE(value < 0 || (value >= PF_NT && value <= PF_NT_DENSE && value % 128 == 0), "calm_hip_configure(\"pf_chunk\"): 1024 ... 2048 in steps of 128");
		slot = &g_pf_chunk;
	} else if (!strcmp(key, "pf_chunk_moe")) {
		CALM_REQUIRE(value < 0 || value == 1024 || value == 2048 || value == PF_NT_MOE, "calm_hip_configure(\"pf_chunk_moe\"): 1024, 2048 or 4096");
		slot = &g_pf_chunk_moe;
	} else if (!strcmp(key, "pf_score_mb")) {
		CALM_REQUIRE(value < 0 || value >= 1, "calm_hip_configure(\"pf_score_mb\"): at least 1 MiB");
		slot = &g_pf_score_mb;
	} else if (!strcmp(key, "stage")) {
		CALM_REQUIRE(value < (int)g_devs.size(), "calm_hip_configure(\"stage\"): no such stage");
		int old_stage = g_alloc_stage;
		g_alloc_stage = value; // (-1 is a value here, not "query")
		return old_stage;
	} else {
		return -1;
	}
	int old = *slot;
	if (value >= 0) {
		if (*slot != value && slot != &g_use_graph && slot != &g_prof) {
			// a knob that shapes the launches: the captured graphs of every prepared model are stale
			for (auto& kv : g_ctx) {
				for (auto& ge : kv.second->graphs) {
					if (ge.second.exec) {
						HIP_CHECK(hipGraphExecDestroy(ge.second.exec));
					}
					if (ge.second.graph) {
						HIP_CHECK(hipGraphDestroy(ge.second.graph));
					}
				}
				kv.second->graphs.clear();
			}
		}
		*slot = value;
	}
	return old;
}

// read-only state (kept apart from the knobs: nothing here changes a launch)
extern "C" int calm_hip_query(const char* key, int value) {
	init_hip();
	HIP_CHECK(hipStreamSynchronize(g_stream));
	if (!strcmp(key, "fused_steps")) {
		return (int)(g_fused_steps & 0x7fffffff);
	} else if (!strcmp(key, "fuse_timeouts")) { // bounded waits of k_qkv_attn that expired, over every prepared model (0 unless a launch lost a producer)
		unsigned total = 0;
		for (auto& kv : g_ctx) {
			unsigned e = 0;
			dev_copy_sync(&e, kv.second->fuse_err, sizeof(e), hipMemcpyDeviceToHost);
			total += e;
		}
		return (int)total;
	} else if (!strcmp(key, "stages")) {
		return (int)g_devs.size();
	} else if (!strcmp(key, "stage_device")) { // the device stage `value` sits on
		CALM_REQUIRE(value >= 0 && value < (int)g_devs.size(), "calm_hip_query(\"stage_device\"): no such stage");
		return g_devs[value].dev;
	} else if (!strcmp(key, "handoffs")) { // hand-off copies timed so far (profiled steps of a model split over stages)
		return (int)g_hop_n;
	} else if (!strcmp(key, "handoff_ns")) { // ... and their average duration
		return g_hop_n ? (int)(g_hop_us * 1e3 / (double)g_hop_n) : 0;
	} else if (!strcmp(key, "pf_redone")) {
		return (int)g_pf_redone; // prompt tokens prefill_hip sent back through the serial path (activations beyond binary16)
	}
	return -1;
}

extern "C" void init_hip(void) {
	if (g_device >= 0) {
		return;
	}
	int n = calm_hip_device_count();
	if (n <= 0) {
		fprintf(stderr, "calm_hip: no HIP device visible -- this backend has no CPU fallback\n");
		abort();
	}
	const char* dv = getenv("CALM_HIP_DEVICE");
	if (!dv) {
		dv = getenv("LOCAL_RANK");
	}
	int dev = dv ? atoi(dv) % n : 0;
	HIP_CHECK(hipSetDevice(dev));
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, dev));
	g_ncu = prop.multiProcessorCount;
	snprintf(g_devname, sizeof(g_devname), "%s", prop.name[0] ? prop.name : prop.gcnArchName);
	HIP_CHECK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
	g_device = dev;
	{
		DevSlot d0;
		d0.dev = dev, d0.stream = g_stream, d0.ncu = g_ncu;
		g_devs.push_back(d0);
		const int stages = env_int("CALM_HIP_DEVICES", 1);
		CALM_REQUIRE(stages >= 1 && stages <= 64, "CALM_HIP_DEVICES must be between 1 and 64");
		for (int s_ = 1; s_ < stages; ++s_) {
			DevSlot d;
			d.dev = (dev + s_) % n;
			HIP_CHECK(hipSetDevice(d.dev));
			hipDeviceProp_t pr;
			HIP_CHECK(hipGetDeviceProperties(&pr, d.dev));
			d.ncu = pr.multiProcessorCount;
			HIP_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
			const int prev = g_devs[s_ - 1].dev;
			if (d.dev != prev) {
				// Adjacent stages on different devices: stage s - 1 PUSHES the residual stream into stage s's memory with
				// hipMemcpyPeerAsync on its own stream (forward_multi), and a tensor a host placed on the other device is pulled with
				// hipMemcpyPeer (upload_pending).  hipMemcpyPeer* is documented to work without peer access (the runtime then stages
				// through host memory) -- which is exactly the silent fallback this path must not take: with access enabled the copy
				// is one DMA over xGMI.  hipDeviceEnablePeerAccess(peer) grants the CURRENT device access to peer's memory and is
				// one-directional, so it is enabled BOTH ways: d -> prev (pulls, and the runtime's choice of the copying engine) and
				// prev -> d (the push).  No silent fallback: refuse devices that cannot reach each other.
				int can_fwd = 0, can_back = 0;
				HIP_CHECK(hipDeviceCanAccessPeer(&can_back, d.dev, prev));
				HIP_CHECK(hipDeviceCanAccessPeer(&can_fwd, prev, d.dev));
				CALM_REQUIRE((can_fwd && can_back) || env_int("CALM_HIP_ALLOW_NO_PEER", 0),
				             "CALM_HIP_DEVICES: adjacent pipeline stages sit on devices without peer access (xGMI / PCIe P2P); set CALM_HIP_ALLOW_NO_PEER=1 to run anyway");
				auto enable = [&](int from, int to, int can) {
					if (!can) {
						return;
					}
					HIP_CHECK(hipSetDevice(from));
					hipError_t e = hipDeviceEnablePeerAccess(to, 0);
					if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
						HIP_CHECK(e);
					}
					(void)hipGetLastError();
				};
				enable(d.dev, prev, can_back);
				enable(prev, d.dev, can_fwd);
				HIP_CHECK(hipSetDevice(d.dev));
			}
			g_devs.push_back(d);
		}
		use_dev(0);
	}
	g_use_graph = env_int("CALM_HIP_GRAPH", 1);
	g_prof = env_int("CALM_HIP_PROF", 0);
	g_prof_json = getenv("CALM_HIP_PROF_JSON");
	if (g_prof_json && *g_prof_json) {
		atexit(write_prof_json);
	} else {
		g_prof_json = nullptr;
	}
	g_split_t = env_int("CALM_HIP_SPLIT_T", g_split_t);
	g_split_min = env_int("CALM_HIP_SPLIT_MIN", g_split_min);
	g_attn_vt = env_int("CALM_HIP_ATTN_VT", g_attn_vt);
	g_qkv_attn = env_int("CALM_HIP_QKV_ATTN", g_qkv_attn);
	g_pf_chunk = env_int("CALM_HIP_PF_CHUNK", g_pf_chunk);
	CALM_REQUIRE(g_pf_chunk >= PF_NT && g_pf_chunk <= PF_NT_DENSE && g_pf_chunk % 128 == 0, "CALM_HIP_PF_CHUNK: 1024 ... 2048 in steps of 128");
	g_pf_chunk_moe = env_int("CALM_HIP_PF_CHUNK_MOE", g_pf_chunk_moe);
	CALM_REQUIRE(g_pf_chunk_moe == 1024 || g_pf_chunk_moe == 2048 || g_pf_chunk_moe == PF_NT_MOE, "CALM_HIP_PF_CHUNK_MOE: 1024, 2048 or 4096");
	g_pf_score_mb = env_int("CALM_HIP_PF_SCORE_MB", g_pf_score_mb);
	CALM_REQUIRE(g_pf_score_mb >= 1, "CALM_HIP_PF_SCORE_MB: at least 1 MiB");
	if (env_int("CALM_HIP_VERBOSE", 0)) {
		printf("# HIP: %s (%s), %d CUs, %.1f GiB, device %d\n", prop.name, prop.gcnArchName, g_ncu, (double)prop.totalGlobalMem / (1024.0 * 1024 * 1024), dev);
	}
}

extern "C" void* upload_hip(void* host, size_t size) {
	init_hip();
	if (g_devs.size() > 1) {
		if (g_alloc_stage < 0) {
			g_pending_uploads[host] = size; // which device it goes to is known in prepare_hip
			return host;
		}
		use_dev(g_alloc_stage);
	}
	void* device = dev_alloc(size);
	staged_upload(device, host, size);
	return device;
}

extern "C" void* alloc_hip(size_t size) {
	init_hip();
	if (g_devs.size() > 1) {
		CALM_REQUIRE(g_alloc_stage >= 0, "alloc_hip with CALM_HIP_DEVICES > 1: say which stage first (calm_hip_configure(\"stage\", s))");
		use_dev(g_alloc_stage);
	}
	return dev_alloc(size);
}

extern "C" void free_hip(void* device) {
	if (!device) {
		return;
	}
	if (g_pending_uploads.erase(device)) {
		return; // a deferred upload that never reached a device: the pointer is the host's own
	}
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, device) != hipSuccess || attr.type != hipMemoryTypeDevice) {
		(void)hipGetLastError();
		return; // (multi-device: upload_hip handed the host pointer back; prepare_hip's device copies are freed by release_hip)
	}
	HIP_CHECK(hipFree(device));
}

extern "C" void download_hip(void* host, const void* device, size_t size) {
	init_hip();
	dev_copy_sync(host, device, size, hipMemcpyDeviceToHost); // behind everything queued on the decode stream
}

namespace {

// one device's share of a model: a whole model (the single-device backend) or one pipeline stage (an ordinary struct
// Transformer holding a contiguous run of layers); everything is allocated on the CURRENT device
void prepare_ctx(struct Transformer* t) {
	struct Config* p = &t->config;
	struct Weights* w = &t->weights;
	struct RunState* s = &t->state;

	Ctx* c = new Ctx();
	c->t = t;
	c->dim = p->dim, c->hidden = p->hidden_dim, c->head_dim = p->head_dim, c->n_layers = p->n_layers;
	c->n_heads = p->n_heads, c->n_kv_heads = p->n_kv_heads, c->vocab = p->vocab_size, c->seq_len = p->seq_len;
	c->q_dim = p->head_dim * p->n_heads, c->kv_dim = p->head_dim * p->n_kv_heads, c->kv_mul = p->n_heads / p->n_kv_heads;
	c->n_experts = p->n_experts, c->n_active = p->n_experts_ac, c->dbits = w->dbits, c->kvbits = s->kvbits;

	CALM_REQUIRE(c->dbits == 4 || c->dbits == 8 || c->dbits == 16, "dbits must be 4, 8 or 16");
	CALM_REQUIRE(c->kvbits == 8 || c->kvbits == 16, "kvbits must be 8 or 16 and set before prepare_hip");
	int G = 128 / c->dbits;
	// every weight row must be a whole number of 16-byte lane-loads (the reference's CPU path asserts
	// n % 16 / n % 32, src/infer.c:47,75,103; its CUDA path dims % 32, src/infer.cu:670)
	CALM_REQUIRE(c->dim % G == 0 && c->hidden % G == 0 && c->q_dim % G == 0, "dim, hidden_dim and n_heads*head_dim must be multiples of 128/dbits");
	CALM_REQUIRE(c->dim % 4 == 0 && c->hidden % 4 == 0 && c->q_dim % 4 == 0 && c->kv_dim % 4 == 0, "dims must be multiples of 4");
	CALM_REQUIRE(c->head_dim % 8 == 0 && c->head_dim <= 512, "head_dim must be a multiple of 8, at most 512");
	CALM_REQUIRE(p->n_layers <= CALM_MAX_LAYERS && p->n_experts <= CALM_MAX_EXPERTS, "too many layers / experts");
	CALM_REQUIRE(p->n_heads % p->n_kv_heads == 0, "n_heads must be a multiple of n_kv_heads");
	CALM_REQUIRE(p->seq_len > CALM_KV_SINKS, "seq_len too small");
	CALM_REQUIRE(!p->n_experts || (p->n_experts_ac > 0 && p->n_experts_ac <= p->n_experts), "bad MoE configuration");
	{
		// the matvec kernels keep their whole input vector in LDS as fp32 (gf4: plus one word sum per 8 columns): dim and
		// n_heads*head_dim must fit the CU's 160 KiB (~40K floats at fp16 / fp8, ~36K at gf4).  hidden_dim need not:
		// k_ffn_down covers a wider one in several launches over column ranges (launch_ffn_down).  Documented in calm_hip.h.
		int widest = c->q_dim > c->dim ? c->q_dim : c->dim;
		size_t need = c->dbits == 16 ? lds_bytes<16>(widest) : (c->dbits == 8 ? lds_bytes<8>(widest) : lds_bytes<4>(widest));
		CALM_REQUIRE(need <= 160 * 1024, "dim / n_heads*head_dim too wide: the activation vector must fit the 160 KiB LDS (about 40K floats)");
	}
	c->lpr = 4;
	while (c->lpr * 8 < c->head_dim) {
		c->lpr *= 2;
	}

	int nact = c->n_active > 0 ? c->n_active : 1;
	c->x = (float*)dev_alloc(c->dim * sizeof(float));
	c->xb = (float*)dev_alloc(c->dim * sizeof(float));
	c->q = (float*)dev_alloc(c->q_dim * sizeof(float));
	c->att = (float*)dev_alloc(c->q_dim * sizeof(float));
	c->he = (float*)dev_alloc((size_t)nact * c->hidden * sizeof(float));
	c->partial = (float*)dev_alloc((size_t)c->n_heads * MAX_SPLIT * (c->head_dim + 4) * sizeof(float));
	c->logits_d = (float*)dev_alloc((size_t)c->vocab * sizeof(float));
	// routing of the last step: [layer][rank] weights, then [layer][rank] expert ids, one allocation (shown to the host as state.exp)
	c->moe_w = (float*)dev_alloc((size_t)c->n_layers * CALM_MAX_EXPERTS * (sizeof(float) + sizeof(int)));
	c->moe_e = (int*)(c->moe_w + (size_t)c->n_layers * CALM_MAX_EXPERTS);
	dev_zero(c->moe_w, (size_t)c->n_layers * CALM_MAX_EXPERTS * (sizeof(float) + sizeof(int)));
	c->next_tok = (int*)dev_alloc(sizeof(int));
	c->sample_st = (SampleState*)dev_alloc(sizeof(SampleState));
	c->trace_count = (int*)dev_alloc(sizeof(int));
	c->trace_cap = 1 << 16;
	c->trace = (int*)dev_alloc((size_t)c->trace_cap * sizeof(int));
	c->ts = (TokState*)dev_alloc(sizeof(TokState));
	dev_zero(c->ts, sizeof(TokState));
	// k_qkv_attn's hand-off granules, tag 0 = never written (TokState::epoch starts at 1), and the word its bounded waits raise
	c->gran = (unsigned long long*)dev_alloc((size_t)(c->q_dim + 2 * c->kv_dim) * sizeof(unsigned long long));
	dev_zero(c->gran, (size_t)(c->q_dim + 2 * c->kv_dim) * sizeof(unsigned long long));
	c->fuse_err = (unsigned*)dev_alloc(sizeof(unsigned));
	dev_zero(c->fuse_err, sizeof(unsigned));
	dev_zero(c->trace_count, sizeof(int));
	dev_zero(c->xb, c->dim * sizeof(float));

	// KV cache, private layout [layer][kv_head][seq_len][head_dim]; zero like calloc (src/infer.c:162-163)
	c->kv_layer_bytes = (size_t)c->kv_dim * c->seq_len * (c->kvbits / 8);
	// head size 128: the value cache a second time behind the first, transposed ([layer][kv_head][head_dim][seq_len]) -- the operand
	// layout of the matrix-core split attention (kernels.hip.h k_attn_vt); both are written by the same epilogues
	const size_t kv_bytes = c->kv_layer_bytes * c->n_layers;
	// ... only where it can be used: the knob "attn_vt" on at prepare time (CALM_HIP_ATTN_VT=0 or calm_hip_configure before
	// prepare_hip saves the memory: + 50 % of the KV cache), whole 64-position blocks (attn_vt_offset), and a window longer than
	// the contexts the unsplit kernel serves.  It is a separate allocation: if it does not fit, the backend runs without it
	// (k_attn_gqa; the epilogues skip the transposed stores when Ctx::vt is null).
	const bool vt = g_attn_vt && attn_has_vt(c->head_dim) && c->seq_len % 64 == 0 && c->seq_len > g_split_min;
	c->kc = dev_alloc(kv_bytes);
	// (V and V^T in one allocation, V^T behind V: the test hooks find it from state.value_cache and the allocation's size)
	if (vt && hipMalloc(&c->vc, 2 * kv_bytes + DEV_PAD) == hipSuccess) {
		c->vt = (char*)c->vc + kv_bytes;
	} else {
		if (vt) {
			(void)hipGetLastError();
			fprintf(stderr, "calm_hip: no room for the transposed value cache (%.1f GiB): split attention falls back to k_attn_gqa\n", (double)kv_bytes / (1 << 30));
		}
		c->vc = dev_alloc(kv_bytes);
		c->vt = nullptr;
	}
	dev_zero(c->kc, kv_bytes);
	dev_zero(c->vc, kv_bytes * (c->vt ? 2 : 1));

	// mixture of experts: the router's table per layer (kernels.hip.h k_attn_out GATE / k_gate_prep) and the partial-sum buffer.  Not for
	// parallel-residual models (their FFN reads the attention norm's output, which k_attn_out does not produce).
	if (c->n_experts > 0 && c->n_experts <= GATE_MAX_E && !p->norm_par) {
		int ep = 1;
		while (ep < c->n_experts) {
			ep *= 2;
		}
		c->gate_ep = ep;
		const size_t per_layer = ((size_t)c->dim + 1) * ep;
		c->gate_mt = (float*)dev_alloc(per_layer * c->n_layers * sizeof(float));
		c->gate_part = (float*)dev_alloc((size_t)(GATE_MAX_E + 2) * GATE_COLS * sizeof(float));
		dev_zero(c->gate_part, (size_t)(GATE_MAX_E + 2) * GATE_COLS * sizeof(float));
		for (int l = 0; l < c->n_layers; ++l) {
			float* mt = c->gate_mt + per_layer * l;
			const dim3 grid(ep + (c->dim + 255) / 256);
			switch (c->dbits) {
			case 16:
				hipLaunchKernelGGL(k_gate_prep<16>, grid, dim3(256), 0, g_stream, mt, w->moegate[l], w->rms_ffn_weight[l], c->dim, c->n_experts, ep);
				break;
			case 8:
				hipLaunchKernelGGL(k_gate_prep<8>, grid, dim3(256), 0, g_stream, mt, w->moegate[l], w->rms_ffn_weight[l], c->dim, c->n_experts, ep);
				break;
			default:
				hipLaunchKernelGGL(k_gate_prep<4>, grid, dim3(256), 0, g_stream, mt, w->moegate[l], w->rms_ffn_weight[l], c->dim, c->n_experts, ep);
			}
		}
		HIP_CHECK(hipGetLastError());
	}

	// RoPE frequencies with the host libm -- the very expression of src/infer.c:226 -- so that
	// pos * freq is bit-identical to the CPU path's; cos/sin of one position step for the sink keys
	int half_hd = c->head_dim / 2;
	std::vector<float> freq(half_hd);
	std::vector<float2> cs1(half_hd);
	for (int i = 0; i < half_hd; ++i) {
		int j_head = 2 * i;
		freq[i] = j_head >= p->rotary_dim ? 0.f : 1.0f / powf(p->rope_theta, (float)j_head / (float)p->rotary_dim);
		float val = 1 * freq[i];
		cs1[i] = make_float2(cosf(val), sinf(val));
	}
	c->rope_freq = (float*)dev_alloc(half_hd * sizeof(float));
	c->rope_cs = (float2*)dev_alloc(half_hd * sizeof(float2));
	c->rope_cs1 = (float2*)dev_alloc(half_hd * sizeof(float2));
	dev_copy_sync(c->rope_freq, freq.data(), half_hd * sizeof(float), hipMemcpyHostToDevice);
	dev_copy_sync(c->rope_cs1, cs1.data(), half_hd * sizeof(float2), hipMemcpyHostToDevice);

	// logits land in pinned host memory: the host sampler reads and overwrites them (src/sampler.c:55)
	HIP_CHECK(hipHostMalloc((void**)&c->logits_h, (size_t)c->vocab * sizeof(float), hipHostMallocDefault));
	memset(c->logits_h, 0, (size_t)c->vocab * sizeof(float));

	c->ba.ts = c->ts;
	c->ba.x = c->x;
	c->ba.embed = w->token_embedding_table;
	c->ba.dim = c->dim;
	c->ba.rope_freq = c->rope_freq;
	c->ba.rope_cs = c->rope_cs;
	c->ba.half_hd = half_hd;
	void* ptrs[13] = {&c->ba.ts,  &c->ba.token, &c->ba.tok_src, &c->ba.pos,       &c->ba.kv_sink, &c->ba.kv_pos, &c->ba.kv_len,
	                  &c->ba.x,   &c->ba.embed, &c->ba.dim,     &c->ba.rope_freq, &c->ba.rope_cs, &c->ba.half_hd};
	memcpy(c->ba_ptrs, ptrs, sizeof(ptrs));

	// what the host program may look at (src/infer.cu:101-112)
	s->x = c->x;
	s->xb = c->xb;
	s->hb = c->he;
	s->he = c->he;
	s->q = c->q;
	s->att = c->att;
	s->exp = c->moe_w; // (extension: the reference's GPU backend leaves it unset) include/calm_hip.h, prepare_hip
	s->key_cache = c->kc;
	s->value_cache = c->vc;
	s->logits = c->logits_h;

	HIP_CHECK(hipDeviceSynchronize()); // uploads done; the host may unmap its copies
	g_ctx[t] = c;
}

// ---- the layer pipeline inside the library (CALM_HIP_DEVICES > 1) --------------------------------------------------
struct MultiCtx {
	std::vector<struct Transformer*> stage; // one trimmed struct Transformer per stage, each prepared on its own device
	std::vector<std::vector<void*>> owned;  // the device copies of each stage's tensors
	std::vector<hipEvent_t> handoff;        // stage s's residual stream has arrived on stage s + 1
	// Stage s + 1 has finished the step (or prompt chunk) whose input it was handed: only then may stage s overwrite that input
	// with the next one.  Steps that return logits end with a host synchronisation and cannot overlap, but FF_UPDATE_KV_ONLY
	// steps -- run.c's prompt loop -- are only enqueued: without this a faster stage s runs ahead into stage s + 1's residual.
	std::vector<hipEvent_t> done;
	std::vector<char> done_valid;
	// CALM_HIP_PROF / knob "prof": an event pair around every hand-off copy (created on first use)
	std::vector<hipEvent_t> hop0, hop1;
};
std::map<struct Transformer*, MultiCtx*> g_multi;

void* upload_pending(const void* host, std::vector<void*>& owned) {
	if (!host) {
		return nullptr;
	}
	auto it = g_pending_uploads.find(host);
	if (it == g_pending_uploads.end()) {
		// placed by the host already (calm_hip_configure("stage", s) before upload_hip / alloc_hip): must be device memory
		hipPointerAttribute_t attr;
		CALM_REQUIRE(hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeDevice,
		             "multi-device: a weight pointer that came neither from a deferred upload_hip nor from a staged upload_hip / alloc_hip");
		int cur = 0;
		HIP_CHECK(hipGetDevice(&cur));
		if (attr.device != cur) {
			// placed on another stage's device (a tied classifier is the embedding table, which lives on stage 0): this stage
			// gets its own copy -- peer access only reaches one stage back, and a classifier read over xGMI every token would
			// be the slowest kernel of the step
			size_t size = 0;
			void* base = nullptr;
			HIP_CHECK(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)host));
			CALM_REQUIRE(base == host && size > DEV_PAD, "multi-device: a weight pointer into the middle of an allocation on another device");
			void* d = dev_alloc(size - DEV_PAD);
			HIP_CHECK(hipMemcpyPeerAsync(d, cur, host, attr.device, size - DEV_PAD, g_stream)); // (the source is complete: uploads return synchronised)
			HIP_CHECK(hipStreamSynchronize(g_stream));
			owned.push_back(d);
			return d;
		}
		return const_cast<void*>(host);
	}
	void* d = dev_alloc(it->second);
	staged_upload(d, host, it->second);
	owned.push_back(d);
	return d;
}

void prepare_multi(struct Transformer* t) {
	const int P = (int)g_devs.size(), L = t->config.n_layers;
	CALM_REQUIRE(L >= P, "CALM_HIP_DEVICES: more pipeline stages than layers");
	MultiCtx* m = new MultiCtx();
	m->owned.resize(P);
	int l0 = 0;
	for (int s = 0; s < P; ++s) {
		const int n = L / P + (s < L % P ? 1 : 0); // contiguous, near-equal runs; the earlier stages take the remainder
		use_dev(s);
		struct Transformer* st = (struct Transformer*)calloc(1, sizeof(struct Transformer));
		st->config = t->config;
		st->config.n_layers = n;
		st->state.kvbits = t->state.kvbits;
		struct Weights *w = &st->weights, *W = &t->weights;
		std::vector<void*>& own = m->owned[s];
		w->dbits = W->dbits;
		const bool first = s == 0, last = s == P - 1;
		if (first) {
			w->token_embedding_table = upload_pending(W->token_embedding_table, own);
		}
		for (int l = 0; l < n; ++l) {
			const int g = l0 + l;
			w->rms_att_weight[l] = (float*)upload_pending(W->rms_att_weight[g], own);
			w->rms_ffn_weight[l] = (float*)upload_pending(W->rms_ffn_weight[g], own);
			w->wq[l] = upload_pending(W->wq[g], own), w->wk[l] = upload_pending(W->wk[g], own), w->wv[l] = upload_pending(W->wv[g], own);
			w->wo[l] = upload_pending(W->wo[g], own);
			w->w1[l] = upload_pending(W->w1[g], own), w->w2[l] = upload_pending(W->w2[g], own), w->w3[l] = upload_pending(W->w3[g], own);
			w->bqkv[l] = (float*)upload_pending(W->bqkv[g], own);
			w->moegate[l] = upload_pending(W->moegate[g], own);
		}
		if (last) {
			w->rms_final_weight = (float*)upload_pending(W->rms_final_weight, own);
			// a tied classifier is the embedding table (src/run.c:112-116): the last stage gets its own copy unless it is also the first
			w->wcls = (W->wcls == W->token_embedding_table && first) ? w->token_embedding_table : upload_pending(W->wcls, own);
		}
		prepare_ctx(st);
		m->stage.push_back(st);
		if (s + 1 < P) {
			hipEvent_t e;
			HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			m->handoff.push_back(e);
		}
		{
			hipEvent_t e;
			HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			m->done.push_back(e);
			m->done_valid.push_back(0);
		}
		l0 += n;
	}
	g_pending_uploads.clear();
	use_dev(0);
	// what the host program may look at: the residual stream where it enters, the logits where they leave
	t->state.x = m->stage[0]->state.x;
	t->state.logits = m->stage[P - 1]->state.logits;
	g_multi[t] = m;
}

// one decode step through every stage: stage s runs on its device's stream, its residual stream crosses to stage s + 1 with a
// peer copy queued behind it, and stage s + 1's stream waits for that copy's event -- the host only waits for the last stage
float* forward_multi(MultiCtx* m, int token, int pos, unsigned flags) {
	const int P = (int)m->stage.size();
	const bool kv_only = (flags & FF_UPDATE_KV_ONLY) != 0;
	for (int s = 0; s < P; ++s) {
		use_dev(s);
		Ctx* c = ctx_of(m->stage[s]);
		if (s > 0) {
			HIP_CHECK(hipStreamWaitEvent(g_stream, m->handoff[s - 1], 0));
		}
		StepPlan sp = {};
		sp.kv_only = kv_only || s + 1 < P; // only the last stage has a final norm and a classifier
		sp.copy_logits = !sp.kv_only;
		run_step(c, token, nullptr, pos, sp, s == 0);
		if (s > 0) {
			HIP_CHECK(hipEventRecord(m->done[s], g_stream)); // this stage is through with the x it was handed
			m->done_valid[s] = 1;
		}
		if (s + 1 < P) {
			Ctx* nx = ctx_of(m->stage[s + 1]);
			if (m->done_valid[s + 1]) {
				HIP_CHECK(hipStreamWaitEvent(g_stream, m->done[s + 1], 0)); // ... of the step before: its x may go now
			}
			const bool timed = g_prof && !kv_only;
			if (timed && m->hop0.empty()) {
				m->hop0.assign(P - 1, nullptr), m->hop1.assign(P - 1, nullptr);
			}
			if (timed && !m->hop0[s]) { // (an event belongs to the device that is current when it is created: stage s's)
				HIP_CHECK(hipEventCreate(&m->hop0[s]));
				HIP_CHECK(hipEventCreate(&m->hop1[s]));
			}
			if (timed) {
				HIP_CHECK(hipEventRecord(m->hop0[s], g_stream));
			}
			HIP_CHECK(hipMemcpyPeerAsync(nx->x, g_devs[s + 1].dev, c->x, g_devs[s].dev, (size_t)c->dim * sizeof(float), g_stream));
			if (timed) {
				HIP_CHECK(hipEventRecord(m->hop1[s], g_stream));
			}
			HIP_CHECK(hipEventRecord(m->handoff[s], g_stream));
		}
	}
	float* out = nullptr;
	if (!kv_only) {
		HIP_CHECK(hipStreamSynchronize(g_stream)); // the last stage's stream: everything before it is ordered by the events
		out = ctx_of(m->stage[P - 1])->logits_h;
		if (g_prof && !m->hop0.empty()) { // the hand-offs of this step (every copy precedes the last stage's work)
			for (int s = 0; s + 1 < P; ++s) {
				float ms = 0;
				HIP_CHECK(hipEventElapsedTime(&ms, m->hop0[s], m->hop1[s]));
				g_hop_us += ms * 1e3;
				g_hop_n++;
			}
		}
	}
	use_dev(0);
	return out;
}

} // namespace

extern "C" void prepare_hip(struct Transformer* t) {
	init_hip();
	if (g_devs.size() > 1) {
		prepare_multi(t);
	} else {
		prepare_ctx(t);
	}
	// the uploads are complete (prepare_ctx drains every device): the pinned staging buffers (2 x 32 MiB per device) go back
	for (auto& kv : g_stagers) {
		HIP_CHECK(hipSetDevice(kv.first));
		for (int b = 0; b < 2; ++b) {
			if (kv.second.pin[b]) {
				HIP_CHECK(hipHostFree(kv.second.pin[b]));
				HIP_CHECK(hipEventDestroy(kv.second.done[b]));
			}
		}
	}
	g_stagers.clear();
	use_dev(0);
}

extern "C" void release_hip(struct Transformer* t) {
	auto mt = g_multi.find(t);
	if (mt != g_multi.end()) {
		MultiCtx* m = mt->second;
		g_multi.erase(mt);
		for (size_t s = 0; s < m->stage.size(); ++s) {
			use_dev((int)s);
			release_hip(m->stage[s]);
			for (void* d : m->owned[s]) {
				HIP_CHECK(hipFree(d));
			}
			free(m->stage[s]);
		}
		for (hipEvent_t e : m->handoff) {
			HIP_CHECK(hipEventDestroy(e));
		}
		for (hipEvent_t e : m->done) {
			HIP_CHECK(hipEventDestroy(e));
		}
		for (size_t i = 0; i < m->hop0.size(); ++i) {
			if (m->hop0[i]) {
				HIP_CHECK(hipEventDestroy(m->hop0[i]));
				HIP_CHECK(hipEventDestroy(m->hop1[i]));
			}
		}
		use_dev(0);
		delete m;
		t->state.x = nullptr, t->state.logits = nullptr;
		return;
	}
	auto it = g_ctx.find(t);
	if (it == g_ctx.end()) {
		return;
	}
	Ctx* c = it->second;
	HIP_CHECK(hipStreamSynchronize(g_stream));
	for (auto& kv : c->graphs) {
		if (kv.second.exec) {
			HIP_CHECK(hipGraphExecDestroy(kv.second.exec));
		}
		if (kv.second.graph) {
			HIP_CHECK(hipGraphDestroy(kv.second.graph));
		}
	}
	for (hipEvent_t e : c->events) {
		HIP_CHECK(hipEventDestroy(e));
	}
	if (c->gate_mt) {
		HIP_CHECK(hipFree(c->gate_mt));
		HIP_CHECK(hipFree(c->gate_part));
	}
	void* bufs[] = {c->x,  c->xb,       c->q,     c->att,         c->he,        c->partial, c->sample_st, c->logits_d, c->moe_w,
	                c->ts, c->next_tok, c->trace, c->trace_count, c->rope_freq, c->rope_cs, c->rope_cs1, c->kc,     c->vc, c->gran, c->fuse_err};
	for (void* b : bufs) {
		HIP_CHECK(hipFree(b));
	}
	void* pf_bufs[] = {c->pf_partial, c->pf_tile_count, c->pf_x, c->pf_xn, c->pf_q, c->pf_att, c->pf_h, c->pf_rope, c->pf_tok, c->pf_gate, c->pf_wsel, c->pf_xe, c->pf_y, c->pf_rows, c->pf_colexp, c->pf_slot, c->pf_logits, c->pf_lp, c->pf_target};
	for (void* b : pf_bufs) {
		if (b) {
			HIP_CHECK(hipFree(b));
		}
	}
	HIP_CHECK(hipHostFree(c->logits_h));
	if (c->pf_flag) {
		HIP_CHECK(hipHostFree(c->pf_flag));
	}
	if (g_prof_ctx == c) {
		g_prof_ctx = nullptr;
	}
	memset(&t->state, 0, offsetof(struct RunState, kvbits));
	t->state.key_cache = t->state.value_cache = nullptr;
	delete c;
	g_ctx.erase(it);
}

extern "C" float* forward_hip(struct Transformer* t, int token, int pos, unsigned flags) {
	auto mt = g_multi.find(t);
	if (mt != g_multi.end()) {
		return forward_multi(mt->second, token, pos, flags);
	}
	Ctx* c = ctx_of(t);
	StepPlan sp = {};
	sp.kv_only = (flags & FF_UPDATE_KV_ONLY) != 0;
	sp.argmax = false;
	sp.copy_logits = !sp.kv_only;
	run_step(c, token, nullptr, pos, sp);
	if (sp.kv_only) {
		return NULL; // enqueued, not synchronised (src/infer.cu:724-727)
	}
	HIP_CHECK(hipStreamSynchronize(g_stream));
	return c->logits_h;
}

extern "C" float* forward_stage_hip(struct Transformer* t, int token, int pos, unsigned flags, unsigned stage_flags) {
	Ctx* c = ctx_of(t);
	const bool first = (stage_flags & CALM_STAGE_FIRST) != 0, last = (stage_flags & CALM_STAGE_LAST) != 0;
	StepPlan sp = {};
	sp.kv_only = !last || (flags & FF_UPDATE_KV_ONLY) != 0;
	sp.argmax = false;
	sp.copy_logits = !sp.kv_only;
	CALM_REQUIRE(sp.kv_only || (t->weights.wcls && t->weights.rms_final_weight), "the last stage needs the final norm and the classifier");
	run_step(c, token, nullptr, pos, sp, first);
	HIP_CHECK(hipStreamSynchronize(g_stream)); // state.x is complete: the host may hand it to the next stage
	return sp.kv_only ? NULL : c->logits_h;
}

extern "C" void copy_hip(void* dst, const void* src, size_t size) {
	init_hip();
	// on the decode stream, then drained: ordered behind the step that produced `src` and ahead of the one that reads `dst`
	// whatever the pointer kinds are (a device-to-device hipMemcpy on the null stream is neither: g_stream is non-blocking
	// and a D2D copy is not host-synchronous)
	HIP_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyDefault, g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream));
}

namespace {

// prefill_hip / prefill_logprobs_hip: logprob == nullptr -> KV cache only.  A model sharded inside the library
// (CALM_HIP_DEVICES) runs a chunk stage after stage: the residual rows X[nb][dim] cross with hipMemcpyPeerAsync + the
// hand-off event, as one token's x does in forward_multi; scoring happens on the last stage.
void prefill_impl(struct Transformer* t, const int* tokens, int n, int pos, float* logprob) {
	std::vector<Ctx*> st;
	MultiCtx* m = nullptr;
	auto mt = g_multi.find(t);
	if (mt != g_multi.end()) {
		m = mt->second;
		for (struct Transformer* stage : m->stage) {
			st.push_back(ctx_of(stage));
		}
	} else {
		st.push_back(ctx_of(t));
	}
	const int P = (int)st.size();
	Ctx* const first = st.front();
	Ctx* const last = st.back();
	auto on_stage = [&](int s) {
		if (m) {
			use_dev(s);
		}
	};
	CALM_REQUIRE(n >= 0 && pos >= 0, "negative token count / position");
	for (int i = 0; i < n; ++i) {
		CALM_REQUIRE(tokens[i] >= 0 && tokens[i] < first->vocab, "token out of range");
	}
	CALM_REQUIRE(!logprob || (last->t->weights.wcls && last->t->weights.rms_final_weight), "scoring needs the final norm and the classifier");
	// The batched path covers the positions before the rolling buffer wraps; positions at or past seq_len
	// (sink rotation between tokens) go through the decode path one token at a time, still on the device.
	// one token through the decode path (positions past the rolling buffer; a chunk whose activations left the binary16 range)
	auto serial_token = [&](int i) {
		const float* l = nullptr;
		if (m) {
			l = forward_multi(m, tokens[i], pos + i, logprob ? 0u : (unsigned)FF_UPDATE_KV_ONLY);
			on_stage(P - 1);
		} else {
			StepPlan sp = {};
			sp.kv_only = logprob == nullptr;
			sp.copy_logits = !sp.kv_only;
			run_step(first, tokens[i], nullptr, pos + i, sp);
			if (logprob) {
				HIP_CHECK(hipStreamSynchronize(g_stream));
				l = first->logits_h;
			}
		}
		if (logprob) {
			float lp = 0.f;
			if (i + 1 < n) { // src/sampler.c:19-32, then the log of src/run.c:298
				float mx = l[0];
				for (int v = 1; v < last->vocab; ++v) {
					mx = l[v] > mx ? l[v] : mx;
				}
				float sum = 0.f;
				for (int v = 0; v < last->vocab; ++v) {
					sum += expf(l[v] - mx);
				}
				lp = (l[tokens[i + 1]] - mx) - logf(sum);
			}
			logprob[i] = lp;
		}
	};
	int done = 0;
	std::vector<int> chunk_start; // first token of every batched chunk of this call
	if ((first->n_experts == 0 || first->n_active <= PF_MAX_ACTIVE) && first->t->weights.token_embedding_table) {
		for (int s = 0; s < P; ++s) {
			on_stage(s);
			pf_alloc(st[s]);
			// the K-range GEMMs leave their tile counters at zero; a launch that did not run to its end must not poison the next call
			HIP_CHECK(hipMemsetAsync(st[s]->pf_tile_count, 0, PF_SPLIT_TILES * sizeof(unsigned), g_stream));
			memset(st[s]->pf_flag, 0, PF_FLAG_WORDS * sizeof(unsigned)); // (the call before ended synchronised: nobody is writing them)
		}
		if (logprob && !last->pf_logits) { // (the last stage's device is current)
			// logits scratch of the scoring GEMM: at most "pf_score_mb" = 256 MiB (2048 tokens at a 32k vocabulary, 512 at 128k, 256 at 256k), whole
			// 128-token columns; prefill_chunk scores a chunk in blocks of that many tokens
			int snt = (int)(((size_t)g_pf_score_mb << 20) / ((size_t)last->vocab * sizeof(float))) / 128 * 128;
			snt = snt < 128 ? 128 : snt > last->pf_nt ? last->pf_nt : snt;
			last->pf_score_nt = snt;
			last->pf_logits = (float*)dev_alloc((size_t)snt * last->vocab * sizeof(float));
			last->pf_lp = (float*)dev_alloc(last->pf_nt * sizeof(float));
			last->pf_target = (int*)dev_alloc(last->pf_nt * sizeof(int));
		}
		int NT = first->pf_nt; // (the stages of a pipeline hold layers of ONE model; buffers of an earlier call may be smaller)
		for (int s = 1; s < P; ++s) {
			NT = st[s]->pf_nt < NT ? st[s]->pf_nt : NT;
		}
		while (done < n && pos + done < first->seq_len) {
			int nb = n - done < NT ? n - done : NT;
			if (pos + done + nb > first->seq_len) {
				nb = first->seq_len - (pos + done);
			}
			if (nb <= 2) {
				break; // a chunk costs about three decode steps whatever its size (it streams every weight once, less efficiently)
			}
			int target[PF_NT_MOE > PF_NT_DENSE ? PF_NT_MOE : PF_NT_DENSE];
			const int k = (int)chunk_start.size() < PF_FLAG_WORDS ? (int)chunk_start.size() : PF_FLAG_WORDS - 1;
			chunk_start.push_back(done);
			for (int s = 0; s < P; ++s) {
				on_stage(s);
				Ctx* c = st[s];
				// this chunk's range flag of this model is the one this device's prompt kernels raise from here on (ordered on the stream)
				HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(calm_pf_range_ptr), &c->pf_flag_ptr[k], sizeof(unsigned*), 0, hipMemcpyHostToDevice, g_stream));
				if (s == 0) {
					HIP_CHECK(hipMemcpyAsync(c->pf_tok, tokens + done, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, g_stream));
				} else {
					HIP_CHECK(hipStreamWaitEvent(g_stream, m->handoff[s - 1], 0));
				}
				const bool score = logprob != nullptr && s == P - 1;
				if (score) {
					for (int b = 0; b < nb; ++b) {
						target[b] = done + b + 1 < n ? tokens[done + b + 1] : -1;
					}
					HIP_CHECK(hipMemcpyAsync(c->pf_target, target, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, g_stream));
				}
				vt_sync(c, pos + done); // the prompt kernels read (and extend) the transposed value cache: rows before the chunk first
				dispatch_prefill_chunk(c, nb, pos + done, score, s == 0);
				if (c->vt && c->vt_rows < pos + done + nb) {
					c->vt_rows = pos + done + nb;
				}
				if (s > 0) {
					HIP_CHECK(hipEventRecord(m->done[s], g_stream));
					m->done_valid[s] = 1;
				}
				if (s + 1 < P) {
					if (m->done_valid[s + 1]) { // stage s + 1 is through with the chunk (or step) before
						HIP_CHECK(hipStreamWaitEvent(g_stream, m->done[s + 1], 0));
					}
					HIP_CHECK(hipMemcpyPeerAsync(st[s + 1]->pf_x, g_devs[s + 1].dev, c->pf_x, g_devs[s].dev, (size_t)nb * c->dim * sizeof(float), g_stream));
					HIP_CHECK(hipEventRecord(m->handoff[s], g_stream));
				}
			}
			if (logprob) { // (the last stage's stream; `target` lives on this stack frame)
				HIP_CHECK(hipMemcpyAsync(logprob + done, last->pf_lp, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, g_stream));
				HIP_CHECK(hipStreamSynchronize(g_stream));
			}
			done += nb;
		}
	}
	if (done > 0) {
		// Did every activation of the batched chunks fit the hi + lo binary16 form (prefill.hip.h: pf_split2)?  The flags are host
		// words the kernels raise: ONE synchronisation per call (of the last stage, which is behind every hand-off), not one
		// per chunk and stage.  If a flag is up -- values beyond +-65504, NaN -- the batched part of the prompt FROM THE FIRST CHUNK
		// THAT RAISED ONE (later chunks read its rows; earlier ones are sound) is redone by the serial fp32 decode path, token by token,
		// before anything is built on it: the batched path never decides a result it cannot represent.  (Its positions lie before the
		// rolling buffer wraps: a serial step there is idempotent, and an unsplit serial step un-mirrors its row of the transposed
		// value cache -- run_step -- so nothing stale of the chunk survives there either.)
		on_stage(P - 1);
		HIP_CHECK(hipStreamSynchronize(g_stream));
		int first_bad = (int)chunk_start.size();
		for (int s = 0; s < P; ++s) {
			for (int k = 0; k < first_bad && k < PF_FLAG_WORDS; ++k) {
				if (st[s]->pf_flag[k] != 0) {
					first_bad = k;
				}
			}
		}
		if (first_bad < (int)chunk_start.size()) {
			const int from = chunk_start[first_bad];
			g_pf_redone += done - from;
			for (int i = from; i < done; ++i) {
				serial_token(i);
			}
		}
	}
	for (; done < n; ++done) {
		serial_token(done);
	}
	// `tokens` may be reused by the caller; KV rows are complete: the last stage's stream is behind every hand-off
	on_stage(P - 1);
	HIP_CHECK(hipStreamSynchronize(g_stream));
	on_stage(0);
}

} // namespace

extern "C" void prefill_hip(struct Transformer* t, const int* tokens, int n, int pos) {
	prefill_impl(t, tokens, n, pos, nullptr);
}

extern "C" void prefill_logprobs_hip(struct Transformer* t, const int* tokens, int n, int pos, float* logprob) {
	CALM_REQUIRE(logprob != nullptr, "logprob buffer missing");
	prefill_impl(t, tokens, n, pos, logprob);
}

extern "C" float* decode_greedy_hip(struct Transformer* t, int token, int pos, int n_steps, int* out_tokens) {
	Ctx* c = ctx_of(t);
	CALM_REQUIRE(n_steps > 0 && n_steps <= c->trace_cap, "n_steps out of range");
	HIP_CHECK(hipMemsetAsync(c->trace_count, 0, sizeof(int), g_stream));
	for (int i = 0; i < n_steps; ++i) {
		StepPlan sp = {};
		sp.argmax = true;
		sp.copy_logits = (i == n_steps - 1);
		run_step(c, i == 0 ? token : 0, i == 0 ? nullptr : c->next_tok, pos + i, sp);
	}
	HIP_CHECK(hipMemcpyAsync(out_tokens, c->trace, (size_t)n_steps * sizeof(int), hipMemcpyDeviceToHost, g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream));
	return c->logits_h;
}

extern "C" float* decode_sample_hip(struct Transformer* t, int token, int pos, int n_steps, int* out_tokens, struct Sampler* sampler) {
	CALM_REQUIRE(sampler, "decode_sample_hip: no sampler");
	if (sampler->temperature == 0.0f || sampler->minp >= 1.0f) { // src/sampler.c:81-83: greedy, no coin drawn
		return decode_greedy_hip(t, token, pos, n_steps, out_tokens);
	}
	Ctx* c = ctx_of(t);
	CALM_REQUIRE(n_steps > 0 && n_steps <= c->trace_cap, "n_steps out of range");
	CALM_REQUIRE(sampler->vocab_size == c->vocab, "decode_sample_hip: the sampler's vocab_size is not the model's");
	SampleState st;
	st.rng = sampler->rng_state;
	st.temperature = sampler->temperature;
	st.cutoff_offset = logf(sampler->minp) * sampler->temperature; // src/sampler.c:52
	HIP_CHECK(hipMemcpyAsync(c->sample_st, &st, sizeof(st), hipMemcpyHostToDevice, g_stream));
	HIP_CHECK(hipMemsetAsync(c->trace_count, 0, sizeof(int), g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream)); // `st` is a stack object
	for (int i = 0; i < n_steps; ++i) {
		StepPlan sp = {};
		sp.sample = true;
		sp.copy_logits = (i == n_steps - 1);
		run_step(c, i == 0 ? token : 0, i == 0 ? nullptr : c->next_tok, pos + i, sp);
	}
	HIP_CHECK(hipMemcpyAsync(out_tokens, c->trace, (size_t)n_steps * sizeof(int), hipMemcpyDeviceToHost, g_stream));
	HIP_CHECK(hipMemcpyAsync(&st, c->sample_st, sizeof(st), hipMemcpyDeviceToHost, g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream));
	sampler->rng_state = st.rng;
	return c->logits_h;
}

extern "C" void perf_hip(void) {
	if (g_hop_n) {
		printf("\nlayer pipeline: %llu hand-offs of the residual stream between stages, avg %.1f usec each (event pair around the peer copy)\n", (unsigned long long)g_hop_n,
		       g_hop_us / (double)g_hop_n);
	}
	Ctx* c = g_prof_ctx;
	if (!c) {
		return;
	}
	static const char* names[CALM_STAGE_COUNT] = {"matmul_qkv", "attn", "matmul_attn", "matmul_ffn_up", "matmul_ffn_down", "output"};
	double total = 0;
	uint64_t runs = 0;
	for (int s = 0; s < CALM_STAGE_COUNT; ++s) {
		total += c->prof[s].us;
	}
	runs = c->prof[CALM_STAGE_QKV].runs / (c->n_layers ? c->n_layers : 1);
	if (!runs) {
		return;
	}
	printf("\nforward_hip breakdown (over %llu runs, avg %.1f usec/run, event-timed eager launches):\n", (unsigned long long)runs, total / runs);
	for (int s = 0; s < CALM_STAGE_COUNT; ++s) {
		if (!c->prof[s].runs) {
			continue;
		}
		printf("\t[%d] %16s: %4.1f%%; %8.1f usec/run, %7.1f GB/s\n", s, names[s], c->prof[s].us / total * 100, c->prof[s].us / runs,
		       (double)c->prof[s].bytes / 1e9 / (c->prof[s].us / 1e6));
	}
}

extern "C" double perf_stage_hip(struct Transformer* t, int stage, int iters, uint64_t* bytes_per_launch) {
	Ctx* c = ctx_of(t);
	CALM_REQUIRE(stage >= 0 && stage < CALM_STAGE_COUNT && iters > 0, "bad stage / iters");
	TokState ts;
	dev_copy_sync(&ts, c->ts, sizeof(ts), hipMemcpyDeviceToHost);
	int kv_len = ts.kv_len > 0 ? ts.kv_len : 1;
	if (bytes_per_launch) {
		*bytes_per_launch = stage_bytes(c, stage, kv_len);
	}
	int n_split = attn_splits(c, kv_len);
	c->attn_chunk = (kv_len + n_split - 1) / n_split;
	c->write_vt = c->vt && n_split > 1;
	// the QKV stage of a step whose attention rides in its launch IS that launch (both stages' bytes); every launch gets a tag of its
	// own (no begin-token kernel between them to count the step up), or the attention would find the previous launch's values ready
	const bool fused = fused_step(c, kv_len, n_split);
	if (fused && stage == CALM_STAGE_QKV && bytes_per_launch) {
		*bytes_per_launch += stage_bytes(c, CALM_STAGE_ATTN, kv_len);
	}
	auto one = [&](int l) {
#define ST(db, kvb)                                  \
	if (c->dbits == db && c->kvbits == kvb) {        \
		switch (stage) {                             \
		case CALM_STAGE_QKV:                         \
			if (fused) {                             \
				c->fuse_salt++;                      \
				launch_qkv_attn<db, kvb>(c, l);      \
			} else {                                 \
				launch_qkv<db, kvb>(c, l);           \
			}                                        \
			break;                                   \
		case CALM_STAGE_ATTN:                        \
			launch_attn<kvb>(c, l, n_split);         \
			break;                                   \
		case CALM_STAGE_ATTN_OUT:                    \
			launch_attn_out<db>(c, l);               \
			break;                                   \
		case CALM_STAGE_FFN_UP:                      \
			launch_ffn_up<db>(c, l);                 \
			break;                                   \
		case CALM_STAGE_FFN_DOWN:                    \
			launch_ffn_down<db>(c, l);               \
			break;                                   \
		case CALM_STAGE_OUTPUT:                      \
			launch_output<db>(c);                    \
			break;                                   \
		}                                            \
	}
		ST(16, 16) ST(8, 16) ST(4, 16) ST(16, 8) ST(8, 8) ST(4, 8)
#undef ST
	};
	hipEvent_t e0, e1;
	HIP_CHECK(hipEventCreate(&e0));
	HIP_CHECK(hipEventCreate(&e1));
	for (int l = 0; l < c->n_layers; ++l) { // warm-up sweep
		one(l);
	}
	HIP_CHECK(hipEventRecord(e0, g_stream));
	for (int it = 0; it < iters; ++it) {
		for (int l = 0; l < c->n_layers; ++l) {
			one(l);
		}
	}
	HIP_CHECK(hipEventRecord(e1, g_stream));
	HIP_CHECK(hipEventSynchronize(e1));
	HIP_CHECK(hipGetLastError());
	float ms = 0;
	HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	HIP_CHECK(hipEventDestroy(e0));
	HIP_CHECK(hipEventDestroy(e1));
	if (c->fuse_salt) { // the tags those launches used are spent: the step count moves past them (graphs are captured with salt 0)
		hipLaunchKernelGGL(k_add_epoch, dim3(1), dim3(1), 0, g_stream, c->ts, c->fuse_salt);
		HIP_CHECK(hipStreamSynchronize(g_stream));
		c->fuse_salt = 0;
	}
	return (double)ms * 1e3 / ((double)iters * c->n_layers);
}

#ifdef CALM_TIMELINE
// tools/timeline.py only (a separate build with -DCALM_TIMELINE): arm / read the per-wave stamps of kernels.hip.h
extern "C" void calm_tl_arm(int waves) {
	init_hip();
	HIP_CHECK(hipStreamSynchronize(g_stream));
	static unsigned long long* buf = nullptr;
	static size_t cap = 0;
	const size_t need = (size_t)waves * 8 * sizeof(unsigned long long);
	if (need > cap) {
		if (buf) {
			HIP_CHECK(hipFree(buf));
		}
		HIP_CHECK(hipMalloc(&buf, need));
		cap = need;
	}
	dev_zero(buf, need);
	unsigned uw = (unsigned)waves;
	HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(calm_tl_buf), &buf, sizeof(buf), 0, hipMemcpyHostToDevice, g_stream));
	HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(calm_tl_waves), &uw, sizeof(uw), 0, hipMemcpyHostToDevice, g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream)); // (stack sources)
}
extern "C" void calm_tl_read(unsigned long long* host, int waves) {
	unsigned long long* buf = nullptr;
	HIP_CHECK(hipMemcpyFromSymbolAsync(&buf, HIP_SYMBOL(calm_tl_buf), sizeof(buf), 0, hipMemcpyDeviceToHost, g_stream));
	HIP_CHECK(hipStreamSynchronize(g_stream));
	dev_copy_sync(host, buf, (size_t)waves * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif
