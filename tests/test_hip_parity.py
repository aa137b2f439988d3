"""Parity of the HIP path against the oracle -- the GPU tests proper.  Everything goes through the
C ABI of libcalm_hip.so (ctypes); the oracle (oracle/) is only the checker.

Tolerances (stated once, used everywhere):
  KERNEL_TOL 2e-5  single kernels: fp32 tree sums vs the oracle's sequential fp32 sums
  LOGIT_TOL  1e-3  whole decode steps, max|delta| / max|logit| per token -- the north-star bound for
                   fp16; fp8 and gf4 WEIGHTS decode exactly and activations stay fp32, so the same bound is
                   the stated tolerance for them too
  FP8KV_TOL  4e-3  the same with an fp8 (e5m2) KV cache on both sides, shallow models from position 0 (measured
                   bound, and why it is not depth-independent: conftest.py, profiles/r05_fp8kv.txt)
(LOGIT_TOL / FP8KV_TOL / logit_tol(kvbits) live in conftest.py: one definition for every test file.)
Integer / index results (argmax, routed experts, greedy token streams) must be identical.
"""
import os

import numpy as np
import pytest

from calm_amd import abi
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel, argmax_first, fptr, generate
from conftest import FP8KV_TOL, GOLDEN_CASES, LOGIT_TOL, load_golden, logit_tol, rel_err
from oracle import oracle

pytestmark = pytest.mark.gpu

KERNEL_TOL = 2e-5


def _rand_w(rng, d, n, dtype, sigma=None):
    w = (rng.standard_normal((d, n)).astype(np.float32) * (sigma or n ** -0.5)).astype(np.float16).astype(np.float32)
    return np.ascontiguousarray(cf.quantize(w, dtype))


# ---------------------------------------------------------------- single kernels ---------------

def test_fp8_decode_is_exact_for_all_256_codes(hiplib):
    """row i holds code i in column 0: W . e0 returns every code's decoded value"""
    w = np.zeros((256, 16), dtype=np.uint8)
    w[:, 0] = np.arange(256)
    x = np.zeros(16, dtype=np.float32)
    x[0] = 1.0
    out = np.empty(256, dtype=np.float32)
    hiplib.calm_hip_test_matvec(8, w.ctypes.data, fptr(x), fptr(out), 16, 256)
    ref = cf.fp8_e5m2_to_f32(np.arange(256, dtype=np.uint8))
    finite = np.isfinite(ref)
    assert np.array_equal(out[finite], ref[finite])
    assert np.array_equal(np.isnan(out), np.isnan(ref)) and np.array_equal(out[np.isinf(ref)], ref[np.isinf(ref)])


def test_gf4_and_fp16_decode_exact(hiplib):
    rng = np.random.default_rng(0)
    words = rng.integers(0, 2**32, size=(64, 4), dtype=np.uint64).astype(np.uint32)
    words = words & 0xFFFFFFBF  # clear the top exponent bit of the e5m2 scale: finite, |scale| < 2
    dec = cf.gf4_to_f32(words.view(np.int32))  # (64, 32)
    halves = rng.integers(0, 2**16, size=(64, 32), dtype=np.uint64).astype(np.uint16)
    halves[(halves & 0x7C00) == 0x7C00] = 0x3C00
    for k in range(32):
        x = np.zeros(32, dtype=np.float32)
        x[k] = 1.0
        out = np.empty(64, dtype=np.float32)
        hiplib.calm_hip_test_matvec(4, words.ctypes.data, fptr(x), fptr(out), 32, 64)
        assert np.array_equal(out, dec[:, k]), k
        hiplib.calm_hip_test_matvec(16, halves.ctypes.data, fptr(x), fptr(out), 32, 64)
        assert np.array_equal(out, halves.view(np.float16)[:, k].astype(np.float32)), k


@pytest.mark.parametrize("dtype", ["fp16", "fp8", "gf4"])
@pytest.mark.parametrize("n,d", [(32, 4), (96, 12), (176, 64), (2048, 256), (4096, 512), (5632, 64), (14336, 32), (6144, 48)])
def test_matvec_matches_oracle(hiplib, dtype, n, d):
    if n % (128 // cf.DBITS[dtype]):
        pytest.skip("row not a whole number of 16-byte pieces for this format")
    rng = np.random.default_rng(n * 7 + d)
    w = _rand_w(rng, d, n, dtype)
    x = rng.standard_normal(n).astype(np.float32)
    out = np.empty(d, dtype=np.float32)
    hiplib.calm_hip_test_matvec(cf.DBITS[dtype], w.ctypes.data, fptr(x), fptr(out), n, d)
    ref = oracle.matvec(w, x, cf.DBITS[dtype], n, d)
    assert rel_err(out, ref) < KERNEL_TOL


def test_matvec_is_linear_at_full_size(hiplib):
    """size-independent property at the BASELINE width: W.(2^k x) == 2^k (W.x) bit for bit"""
    rng = np.random.default_rng(1)
    w = _rand_w(rng, 256, 14336, "fp8")
    x = rng.standard_normal(14336).astype(np.float32)
    a, b = np.empty(256, dtype=np.float32), np.empty(256, dtype=np.float32)
    hiplib.calm_hip_test_matvec(8, w.ctypes.data, fptr(x), fptr(a), 14336, 256)
    x8 = x * np.float32(8)
    hiplib.calm_hip_test_matvec(8, w.ctypes.data, fptr(x8), fptr(b), 14336, 256)
    assert np.array_equal(a * np.float32(8), b)


@pytest.mark.parametrize("dtype", ["fp16", "fp8", "gf4"])
@pytest.mark.parametrize("ln", [0, 1])
@pytest.mark.parametrize("n,d", [(96, 301), (4096, 1000), (6144, 130)])
def test_norm_matvec_matches_oracle(hiplib, dtype, ln, n, d):
    if n % (128 // cf.DBITS[dtype]):
        pytest.skip("row granularity")
    rng = np.random.default_rng(n + d + ln)
    w = _rand_w(rng, d, n, dtype)
    x = (rng.standard_normal(n) * 3 + (0.7 if ln else 0)).astype(np.float32)
    nw = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    out = np.empty(d, dtype=np.float32)
    hiplib.calm_hip_test_norm_matvec(cf.DBITS[dtype], w.ctypes.data, fptr(x), fptr(nw), fptr(out), n, d, 1e-5, ln)
    ref = oracle.matvec(w, oracle.norm(x, nw, 1e-5, bool(ln)), cf.DBITS[dtype], n, d)
    assert rel_err(out, ref) < KERNEL_TOL


@pytest.mark.parametrize("n_heads,n_kv,head_dim", [(4, 2, 16), (3, 3, 32), (32, 4, 64), (32, 8, 128), (6, 1, 96), (8, 8, 256)])
@pytest.mark.parametrize("kv_len,n_split", [(1, 1), (7, 1), (256, 1), (1000, 1), (1000, 4), (37, 5)])
def test_attention_matches_oracle(hiplib, n_heads, n_kv, head_dim, kv_len, n_split):
    rng = np.random.default_rng(n_heads * 1000 + head_dim + kv_len)
    seq_len = max(kv_len, 8)
    kv_dim = n_kv * head_dim
    q = rng.standard_normal(n_heads * head_dim).astype(np.float32)
    k = (rng.standard_normal((seq_len, kv_dim)) * 0.7).astype(np.float16)
    v = rng.standard_normal((seq_len, kv_dim)).astype(np.float16)
    out = np.empty(n_heads * head_dim, dtype=np.float32)
    hiplib.calm_hip_test_attn(fptr(q), k.ctypes.data, v.ctypes.data, fptr(out), n_heads, n_kv, head_dim, seq_len, kv_len, n_split)
    ref = oracle.attention(q, k, v, n_heads, n_kv, head_dim, kv_len)
    assert rel_err(out, ref) < KERNEL_TOL


@pytest.mark.parametrize("n_heads,n_kv", [(32, 8), (8, 8), (6, 1), (2, 1), (48, 8), (7, 1), (16, 2), (24, 2), (5, 1)])
@pytest.mark.parametrize("kv_len,n_split", [(385, 4), (1000, 8), (2049, 32), (4096, 32), (130, 3), (64, 2), (4100, 64)])
def test_split_attention_over_the_transposed_value_cache(hiplib, n_heads, n_kv, kv_len, n_split):
    """head size 128 and a window of whole 64-position blocks: the split kernel is k_attn_vt (matrix cores, V read from the transposed
    copy of the cache; all query heads of a kv head up to 8 per workgroup: DBRX's 6, Yi's 7, 12 as 2 x 6, a prime 5) -- same answer as
    the reference's three loops; splits are rounded up to 64-position blocks, so some of the trailing ones are empty"""
    rng = np.random.default_rng(n_heads * 1000 + kv_len)
    head_dim = 128
    seq_len = (kv_len + 63) // 64 * 64 + 64
    kv_dim = n_kv * head_dim
    q = rng.standard_normal(n_heads * head_dim).astype(np.float32)
    k = (rng.standard_normal((seq_len, kv_dim)) * 0.7).astype(np.float16)
    v = rng.standard_normal((seq_len, kv_dim)).astype(np.float16)
    out = np.empty(n_heads * head_dim, dtype=np.float32)
    hiplib.calm_hip_test_attn(fptr(q), k.ctypes.data, v.ctypes.data, fptr(out), n_heads, n_kv, head_dim, seq_len, kv_len, n_split)
    ref = oracle.attention(q, k, v, n_heads, n_kv, head_dim, kv_len)
    assert rel_err(out, ref) < KERNEL_TOL
    # what the cache holds beyond the live range (slots of an earlier, longer sequence) must not matter -- not even infinities
    k[kv_len:] = np.float16(np.nan)
    v[kv_len:] = np.float16(np.inf)
    out2 = np.empty_like(out)
    hiplib.calm_hip_test_attn(fptr(q), k.ctypes.data, v.ctypes.data, fptr(out2), n_heads, n_kv, head_dim, seq_len, kv_len, n_split)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("scale", [1e5, 3e-6, 1.0])
def test_split_attention_query_range(hiplib, scale):
    """k_attn_vt carries the fp32 query as hi + lo binary16: components beyond 65504 (or far below 1) must not turn the step into
    NaN or lose the low part -- the query is scaled by a power of two per head first.  One head's query is `scale` times the others'"""
    rng = np.random.default_rng(11)
    n_heads, n_kv, head_dim, kv_len, n_split = 8, 2, 128, 1000, 8
    seq_len = 1024
    q = rng.standard_normal(n_heads * head_dim).astype(np.float32)
    q[head_dim:2 * head_dim] *= np.float32(scale)
    q[5 * head_dim] = 0.0
    k = (rng.standard_normal((seq_len, n_kv * head_dim)) * (0.7 / max(scale, 1.0) ** 0.5)).astype(np.float16)
    v = rng.standard_normal((seq_len, n_kv * head_dim)).astype(np.float16)
    out = np.empty(n_heads * head_dim, dtype=np.float32)
    hiplib.calm_hip_test_attn(fptr(q), k.ctypes.data, v.ctypes.data, fptr(out), n_heads, n_kv, head_dim, seq_len, kv_len, n_split)
    ref = oracle.attention(q, k, v, n_heads, n_kv, head_dim, kv_len)
    assert np.isfinite(out).all() and rel_err(out, ref) < 1e-4


def test_attention_softmax_is_stable_for_huge_scores(hiplib):
    """scores of +-1e4: the reference subtracts the max (src/infer.c:252-257); so must we"""
    rng = np.random.default_rng(3)
    q = (rng.standard_normal(4 * 64) * 100).astype(np.float32)
    k = (rng.standard_normal((64, 2 * 64)) * 100).astype(np.float16)
    v = rng.standard_normal((64, 2 * 64)).astype(np.float16)
    out = np.empty(4 * 64, dtype=np.float32)
    hiplib.calm_hip_test_attn(fptr(q), k.ctypes.data, v.ctypes.data, fptr(out), 4, 2, 64, 64, 64, 1)
    ref = oracle.attention(q, k, v, 4, 2, 64, 64)
    assert np.isfinite(out).all() and rel_err(out, ref) < 1e-4


def test_device_argmax_is_first_strict_maximum(hiplib):
    rng = np.random.default_rng(4)
    for n in (5, 1024, 32000, 128256):
        lg = rng.standard_normal(n).astype(np.float32)
        assert hiplib.calm_hip_test_argmax(fptr(lg), n) == oracle.argmax(lg) == int(np.argmax(lg))
        j = int(rng.integers(0, n))
        lg[j] = lg.max()  # an exact tie: the lower index wins
        assert hiplib.calm_hip_test_argmax(fptr(lg), n) == oracle.argmax(lg)
    lg = np.array([np.nan, -3.0, np.nan, -3.0], dtype=np.float32)
    assert hiplib.calm_hip_test_argmax(fptr(lg), 4) == oracle.argmax(lg) == 1
    # nothing to pick (all NaN / all <= -FLT_MAX): the reference returns -1 and would index the embedding table with it;
    # the device sampler feeds its pick straight back into the next step, so it answers 0 there
    for lg in (np.full(7, np.nan, dtype=np.float32), np.full(70, -np.inf, dtype=np.float32)):
        assert oracle.argmax(lg) == -1
        assert hiplib.calm_hip_test_argmax(fptr(lg), lg.size) == 0


# ---------------------------------------------------------------- whole decode steps ------------

@pytest.mark.parametrize("graph", [1, 0])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_golden_logits_teacher_forced(hiplib, case, graph):
    """the reference's own logits (tests/golden, from src/infer.c) reproduced by forward_hip"""
    model, z = load_golden(case)
    hiplib.calm_hip_configure(b"graph", graph)
    b = HipBackend(model)
    try:
        worst = 0.0
        for pos, tok in enumerate(z["tokens"]):
            lg = b.forward(int(tok), pos, 0)
            ref = z["logits"][pos]
            worst = max(worst, rel_err(lg, ref))
            top2 = np.partition(ref, -2)[-2:]
            if top2[1] - top2[0] > 4 * LOGIT_TOL * np.abs(ref).max():
                assert argmax_first(lg) == int(np.argmax(ref)), pos
        assert worst < LOGIT_TOL, worst
        # the KV cache the steps left behind (layer 0), against the reference's cache
        k = b.read_kv(0, 0).astype(np.float32)
        v = b.read_kv(0, 1).astype(np.float32)
        kg = z["k_last"].view(np.float16).astype(np.float32)
        vg = z["v_last"].view(np.float16).astype(np.float32)
        assert np.abs(k - kg).max() <= 2e-3 * max(np.abs(kg).max(), 1.0)
        assert np.abs(v - vg).max() <= 2e-3 * max(np.abs(vg).max(), 1.0)
    finally:
        b.close()
        hiplib.calm_hip_configure(b"graph", 1)


@pytest.mark.parametrize("route", [1, 0])
@pytest.mark.parametrize("case", ["moe6_fp8", "moe12_ln_fp16", "dbrx_like_fp8"])
def test_expert_counts_that_are_not_powers_of_two_route_like_the_reference(hiplib, case, route):
    """6 experts top-2 (RMSNorm) and 12 experts top-3 (LayerNorm): k_attn_out<GATE> pads its expert rows to a power of two and files the
    norm statistics behind them; k_ffn_up<MOE = 2> must read them THERE (round 4 read rows n_experts, n_experts + 1: the padded experts'
    zeros, so the picks were right and the mixture weights silently wrong).  Both routing forms -- ahead, from the
    partial sums (the rule); "forms" 8: the gate inside k_ffn_up -- against the reference's logits, and the routed experts of the last step against the
    oracle's (src/infer.c:277-305)."""
    model, z = load_golden(case)
    old = hiplib.calm_hip_configure(b"forms", 0 if route else 8)
    b = HipBackend(model)
    cpu = oracle.OracleBackend(model)
    try:
        L, k = model.config.n_layers, model.config.n_experts_ac
        for pos, tok in enumerate(z["tokens"]):
            lg = b.forward(int(tok), pos, 0)
            _, eo, wo, _ = cpu.forward_traced(int(tok), pos)
            assert rel_err(lg, z["logits"][pos]) < LOGIT_TOL, pos
            for layer in range(L):
                e, w = b.read_moe(layer)
                assert list(e[:k]) == list(eo[layer][:k]), (pos, layer, e, eo[layer])
                assert np.abs(np.asarray(w[:k]) - np.asarray(wo[layer][:k])).max() < 1e-4, (pos, layer, w, wo[layer])
                assert abs(float(np.sum(w[:k])) - 1.0) < 1e-5
    finally:
        b.close()
        hiplib.calm_hip_configure(b"forms", old)


@pytest.mark.parametrize("case", ["tiny_fp8", "moe_fp8", "sink_fp16", "bias_tied_gf4", "moe_gf4", "dbrx_like_fp8"])
def test_alternative_tile_shapes_give_the_reference_logits(hiplib, case):
    """the tile shapes the launchers pick by matrix size (k_qkv half-depth tiles for small matrices, one row per task in k_attn_out /
    k_ffn_down when row pairs would leave a grid's last round half empty, gf4's old 2 x 7 shape) forced on: same logits"""
    if case not in GOLDEN_CASES:
        pytest.skip("no such golden case")
    model, z = load_golden(case)
    assert hiplib.calm_hip_configure(b"forms", 2) == 0  # 0 = "by the rule" is the default; 2: the small-matrix forms always
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(z["tokens"]):
            assert rel_err(b.forward(int(tok), pos, 0), z["logits"][pos]) < LOGIT_TOL, pos
    finally:
        b.close()
        hiplib.calm_hip_configure(b"forms", 0)


@pytest.mark.parametrize("name,dtype,layers", [("mistral-7b", "fp8", 2), ("llama-3-8b", "gf4", 1), ("tinyllama-1.1b", "fp16", 2), ("mixtral-8x7b", "fp8", 1)])
def test_launch_forms_that_only_reorder_work_are_bit_identical(hiplib, name, dtype, layers):
    """Round 4's forms that change WHO does a row or WHERE its activations are read from, not its arithmetic -- activations in registers,
    the skewed deal of k_ffn_up's task rounds, all experts' images side by side in k_ffn_down -- must leave every logit bit-identical
    to the plain forms ("forms" 1; full-width shapes: the forms only engage at dim 4096 / 2048 and full grids).
    "forms" 8 (the gate computed inside k_ffn_up) changes the router's summation order: same experts away from near-ties, logits within the common tolerance."""
    spec = cf.SPECS[name]
    tensors, md = cf.synth_model_big(spec, dtype, seed=17, n_layers=layers)
    model = HostModel(tensors, md, context=64)
    toks = [11, 4242, 7, 31000]

    def run():
        b = HipBackend(model)
        try:
            return np.stack([b.forward(t, pos, 0).copy() for pos, t in enumerate(toks)])
        finally:
            b.close()

    base = run()
    old = hiplib.calm_hip_configure(b"forms", 1)
    try:
        assert np.array_equal(run(), base)
    finally:
        hiplib.calm_hip_configure(b"forms", old)
    if spec.n_experts:
        old = hiplib.calm_hip_configure(b"forms", 8)
        try:
            assert rel_err(run(), base) < LOGIT_TOL
        finally:
            hiplib.calm_hip_configure(b"forms", old)


@pytest.mark.parametrize("knob", [1, 0])
@pytest.mark.parametrize("case", ["fuse_hd64_bias_fp16", "fuse_hd128_sink_fp16"])
def test_attention_inside_the_qkv_launch_gives_the_reference_logits(hiplib, case, knob):
    """k_qkv_attn (round 6): short-context attention as the first n_heads workgroups of k_qkv's launch -- cached rows prefetched into
    registers, q / k / v of the token handed over as tagged granules -- against the reference's own logits, and the two-launch form on
    the same fixtures.  The fixtures are the shapes the fused launch takes (heads of 64 / 128, whole-KiB rows); two layers exercise
    the tag's layer field, the rolling 16-row buffer a masked slot in the middle of the cached range and the re-rotated sink keys."""
    model, z = load_golden(case)
    old = hiplib.calm_hip_configure(b"qkv_attn", knob)
    before = hiplib.calm_hip_query(b"fused_steps", 0)
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(z["tokens"]):
            assert rel_err(b.forward(int(tok), pos, 0), z["logits"][pos]) < LOGIT_TOL, pos
        ran = hiplib.calm_hip_query(b"fused_steps", 0) - before
        assert ran == (len(z["tokens"]) if knob else 0), "the step did not take the launch form this test is about"
        assert hiplib.calm_hip_query(b"fuse_timeouts", 0) == 0
        k = b.read_kv(0, 0).astype(np.float32)
        kg = z["k_last"].view(np.float16).astype(np.float32)
        assert np.abs(k - kg).max() <= 2e-3 * max(np.abs(kg).max(), 1.0)
    finally:
        b.close()
        hiplib.calm_hip_configure(b"qkv_attn", old)


@pytest.mark.parametrize("name,dtype,layers,kvbits", [("mistral-7b", "fp8", 2, 16), ("mistral-7b", "fp8", 2, 8), ("tinyllama-1.1b", "fp16", 3, 16), ("llama-3-8b", "gf4", 1, 16)])
def test_attention_inside_the_qkv_launch_agrees_with_two_launches_at_full_width(hiplib, name, dtype, layers, kvbits):
    """the same decode, 300 positions from 0, with the knob on and off at the BASELINE widths: the fused launch serves positions up to
    its register capacity (256 cached rows at head size 128) and hands over to k_qkv + k_attn beyond it in the same sequence.  Logits
    agree to summation order (fp16 cache) / to the e5m2 cache's code flips (FP8KV_TOL, tests/conftest.py); gf4 weights never take the
    fused launch (their k_qkv prefers the grid it cannot give them: launch rule fused_ok) -- the knob must change nothing there."""
    spec = cf.SPECS[name]
    tensors, md = cf.synth_model_big(spec, dtype, seed=23, n_layers=layers)
    model = HostModel(tensors, md, context=512)
    n = 300

    def run(v):
        old = hiplib.calm_hip_configure(b"qkv_attn", v)
        before = hiplib.calm_hip_query(b"fused_steps", 0)
        b = HipBackend(model, kvbits=kvbits)
        try:
            out = np.stack([b.forward((7 * pos + 3) % spec.vocab_size, pos, 0).copy() for pos in range(n)])
            return out, hiplib.calm_hip_query(b"fused_steps", 0) - before
        finally:
            b.close()
            hiplib.calm_hip_configure(b"qkv_attn", old)

    two, ran0 = run(0)
    one, ran1 = run(1)
    assert ran0 == 0 and ran1 == (0 if dtype == "gf4" else (256 if spec.head_dim == 128 else n)), (ran0, ran1)
    assert np.isfinite(one).all() and hiplib.calm_hip_query(b"fuse_timeouts", 0) == 0
    worst = max(rel_err(one[p], two[p]) for p in range(n))
    assert worst < (2e-5 if kvbits == 16 else FP8KV_TOL), worst


@pytest.mark.parametrize("case", ["tiny_fp8", "moe_fp8", "sink_fp16", "bias_tied_gf4"])
def test_greedy_stream_identical_and_device_decode_agrees(hiplib, case):
    """free-running greedy decode: forward_hip + host argmax, generate(), and the device-side
    decode_greedy_hip all produce the reference's token stream"""
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    b = HipBackend(model)
    try:
        out, stats = generate(b, model, [toks[0]], len(toks))
        assert out[:-1] == toks[1:]
        dev, last_logits = b.decode_greedy(toks[0], 0, len(toks))
        assert list(dev[:-1]) == toks[1:]
        assert rel_err(last_logits, z["logits"][-1]) < LOGIT_TOL
    finally:
        b.close()


def test_kv_only_prompt_steps_then_logits(hiplib):
    model, z = load_golden("tiny_fp16")
    toks = [int(t) for t in z["tokens"]]
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(toks[:10]):
            assert b.forward(tok, pos, abi.FF_UPDATE_KV_ONLY) is None
        lg = b.forward(toks[10], 10, 0)
        assert rel_err(lg, z["logits"][10]) < LOGIT_TOL
        # positions may go backwards (perplexity mode resets pos, src/run.c:296): replay from 0
        for pos, tok in enumerate(toks[:5]):
            lg = b.forward(tok, pos, 0)
        assert rel_err(lg, z["logits"][4]) < LOGIT_TOL
    finally:
        b.close()


def test_logits_buffer_is_host_writable_and_steps_are_deterministic(hiplib):
    model, z = load_golden("tiny_fp8")
    b = HipBackend(model)
    try:
        a1 = b.forward(5, 0, 0).copy()
        lg = b.forward(5, 0, 0)
        lg[:] = 0  # the sampler overwrites logits in place (src/sampler.c:55)
        a2 = b.forward(5, 0, 0).copy()
        assert np.array_equal(a1, a2)
    finally:
        b.close()


def test_attention_split_path_inside_forward(hiplib):
    """force the split-KV attention (+ merge kernel) inside whole steps and compare with unsplit"""
    model, z = load_golden("tiny_fp16")
    toks = [int(t) for t in z["tokens"]]
    old = hiplib.calm_hip_configure(b"split_t", 4)
    old_min = hiplib.calm_hip_configure(b"split_min", 2)  # split from the third position on
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(toks):
            lg = b.forward(tok, pos, 0)
            assert rel_err(lg, z["logits"][pos]) < LOGIT_TOL
    finally:
        b.close()
        hiplib.calm_hip_configure(b"split_t", old)
        hiplib.calm_hip_configure(b"split_min", old_min)


@pytest.mark.parametrize("name,dtype,layers", [("mistral-7b", "fp8", 2), ("llama-3-8b", "gf4", 1), ("tinyllama-1.1b", "fp16", 2), ("mixtral-8x7b", "fp8", 1), ("dbrx-132b", "fp8", 1)])
def test_full_width_layer_reduced_models_match_oracle(hiplib, name, dtype, layers):
    """BASELINE.json shapes at full width and vocabulary, depth cut so the CPU oracle takes seconds"""
    spec = cf.SPECS[name]
    tensors, md = cf.synth_model_big(spec, dtype, seed=3, n_layers=layers)
    model = HostModel(tensors, md, context=64)
    o = oracle.OracleBackend(model)
    b = HipBackend(model)
    try:
        tok = 11
        for pos in range(6):
            lo = o.forward(tok, pos, 0)
            lg = b.forward(tok, pos, 0)
            assert rel_err(lg, lo) < LOGIT_TOL, (pos, rel_err(lg, lo))
            tok = oracle.argmax(lo)
    finally:
        b.close()
        o.close()


@pytest.mark.parametrize("dtype", ["fp8", "fp16", "gf4"])
def test_hidden_dim_wider_than_the_lds_runs_as_column_ranges(hiplib, dtype):
    """hidden_dim 49152 (Qwen1.5-72B's): its fp32 image (192 KiB) does not fit a CU's 160 KiB of LDS, so the down-projection runs
    as two launches over column ranges, each adding onto the residual -- against the oracle, MoE variant included"""
    for experts in (0, 4):
        spec = cf.tiny_spec(f"wide_{dtype}_{experts}", hidden_dim=49152, n_layers=1, n_experts=experts, n_experts_active=2 if experts else 0)
        tensors, md = cf.synth_model(spec, dtype, seed=12 + experts)
        model = HostModel(tensors, md)
        o, b = oracle.OracleBackend(model), HipBackend(model)
        try:
            tok = 5
            for pos in range(6):
                lo = o.forward(tok, pos, 0)
                lg = b.forward(tok, pos, 0)
                assert rel_err(lg, lo) < LOGIT_TOL, (dtype, experts, pos, rel_err(lg, lo))
                tok = oracle.argmax(lo)
        finally:
            b.close()
            o.close()


def test_mistral7b_full_depth_properties(hiplib):
    """BASELINE config 2 at full size (32 layers, 7.1 GB fp8, streamed to the GPU): size-independent
    properties -- bitwise determinism, graph == eager, device decode == host-sampled decode,
    pos-rewind idempotence -- plus the reference accounting of the bytes one step reads"""
    spec = cf.SPECS["mistral-7b"]
    model = HostModel(cf.stub_tensors(spec, "fp8"), spec.metadata("fp8"))
    assert model.accounting()[2] == 7_111_458_816
    b = HipBackend(model, stream=cf.synth_stream_big(spec, "fp8", seed=5))
    try:
        toks, _ = generate(b, model, [17], 24)
        ref_last = b.forward(toks[-2], 23, 0).copy()
        dev, last = b.decode_greedy(17, 0, 24)
        assert list(dev) == toks
        assert np.array_equal(last, ref_last)
        hiplib.calm_hip_configure(b"graph", 0)
        toks_eager, _ = generate(b, model, [17], 24)
        hiplib.calm_hip_configure(b"graph", 1)
        assert toks_eager == toks
        assert np.isfinite(ref_last).all() and np.abs(ref_last).max() < 1e4
    finally:
        b.close()


@pytest.mark.skipif(not os.path.exists(oracle.RUN_HIP), reason="oracle/_ref/run_hip (reference CLI linked to libcalm_hip.so) not built")
def test_reference_cli_runs_on_the_hip_backend():
    """the drop-in proof: the reference's UNMODIFIED run.c, bound to libcalm_hip.so by renaming its four
    cuda externs, decodes the same text on the GPU as on its own CPU backend"""
    import subprocess

    from conftest import GOLDEN

    env = dict(os.environ)
    env.pop("CALM_CPU", None)
    r = subprocess.run([oracle.RUN_HIP, os.path.join(GOLDEN, "tiny_fp16.calm"), "-i", "abc abc", "-t", "0", "-n", "32"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.splitlines()[1] + "\n" == open(os.path.join(GOLDEN, "cli_tiny_fp16.txt")).read()
    assert "tok/s" in r.stderr


@pytest.mark.parametrize("case", ["tiny_fp8", "moe_fp8", "bias_tied_gf4"])
def test_pipeline_stages_on_one_gpu_equal_the_unsharded_step(hiplib, case):
    """forward_stage_hip: the model cut into two layer stages (each its own struct Transformer with its own KV
    slice), x handed from stage 0 to stage 1 through copy_hip -- the data path of calm_amd/pipeline.py without
    the transport -- reproduces the reference logits"""
    import ctypes

    from calm_amd.pipeline import stage_model

    model, z = load_golden(case)
    (m0, f0), (m1, f1) = stage_model(model, 0, 2), stage_model(model, 1, 2)
    b0, b1 = HipBackend(m0), HipBackend(m1)
    try:
        hand = np.zeros(model.config.dim, dtype=np.float32)  # a host buffer standing in for the RCCL message
        worst = 0.0
        for pos, tok in enumerate(z["tokens"][:12]):
            assert b0.forward_stage(int(tok), pos, 0, f0) is None
            b0.export_x(hand.ctypes.data)
            b1.import_x(hand.ctypes.data)
            lg = b1.forward_stage(int(tok), pos, 0, f1)
            worst = max(worst, rel_err(lg, z["logits"][pos]))
        assert worst < LOGIT_TOL, worst
    finally:
        b0.close()
        b1.close()


PIPELINE_STAGE_WORKER = """
import json, os, sys
import numpy as np
import torch   # first: the wheel carries its own HIP runtime, which has to come up before libcalm_hip.so's
sys.path.insert(0, os.environ["CALM_ROOT"])
if not torch.cuda.is_available():
    print(json.dumps({"torch_gpu": False})); sys.exit(0)
from calm_amd.host import HipBackend, HostModel, generate
from calm_amd.pipeline import PipelineStage, stage_model


class Loopback:
    # two stages in one process: rank is set per stage, messages are queued tensors (torch.distributed's signatures)
    def __init__(self):
        self.box, self.rank = {}, 0
    def get_rank(self):
        return self.rank
    def get_world_size(self):
        return 2
    def send(self, t, dst):
        self.box[dst] = t.clone()
    def recv(self, t, src):
        t.copy_(self.box.pop(self.rank))
    def broadcast(self, t, src):
        if self.rank == src:
            self.box["tok"] = t.clone()
        else:
            t.copy_(self.box["tok"])


model = HostModel.from_file(os.environ["CALM_MODEL"])
first = int(os.environ["CALM_FIRST"])
whole = HipBackend(model)
want = [int(t) for t in generate(whole, model, [first], 12)[0]]
whole.close()
link = Loopback()
stages = []
for r in range(2):
    sm, flags = stage_model(model, r, 2)
    link.rank = r
    stages.append(PipelineStage(HipBackend(sm), sm.config.dim, flags, link, "cuda"))
tok, got = first, []
for pos in range(12):
    link.rank = 0
    assert stages[0].step(tok, pos) is None
    link.rank = 1
    tok = int(np.argmax(stages[1].step(tok, pos)))
    got.append(tok)
print(json.dumps({"torch_gpu": True, "got": got, "want": want}))
"""


def test_pipeline_stage_class_with_device_tensors(hiplib):
    """calm_amd.pipeline.PipelineStage itself on the GPU: its hand-off buffers are torch DEVICE tensors whose data_ptr() goes
    through copy_hip (import_x / export_x), torch's stream and the backend's being ordered by the synchronisations the class
    places.  One process and one GPU cannot hold two RCCL ranks, so the transport is a loop-back object with torch.distributed's
    send / recv / broadcast signatures (the real rendezvous is covered over gloo in tests/test_pipeline_gloo.py): what is under
    test is the device-pointer path the CPU tests cannot reach.  Greedy stream == the unsharded backend's.  Its own process:
    torch's bundled HIP runtime and the system one libcalm_hip.so links must come up in that order."""
    import json
    import subprocess
    import sys

    from conftest import GOLDEN, ROOT

    model, z = load_golden("moe_fp8")
    env = dict(os.environ, CALM_ROOT=ROOT, CALM_MODEL=os.path.join(GOLDEN, "moe_fp8.calm"), CALM_FIRST=str(int(z["tokens"][0])))
    r = subprocess.run([sys.executable, "-c", PIPELINE_STAGE_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    if not out["torch_gpu"]:
        pytest.skip("this box's torch build sees no GPU (torch.cuda.is_available() is False in a fresh process)")
    assert out["got"] == out["want"]


@pytest.mark.skipif(not os.path.exists(oracle.RUN_HIP), reason="oracle/_ref/run_hip not built")
def test_reference_cli_perplexity_mode_on_the_hip_backend():
    """`run -x file` (study(), src/run.c:258-316) through the unmodified reference CLI on the HIP backend:
    every step returns logits, positions wrap around (pos = i % steps), the host computes log-probs from
    them; the perplexity must equal the reference CPU backend's (golden) to 4 significant digits"""
    import re
    import subprocess

    from conftest import GOLDEN

    env = dict(os.environ)
    env.pop("CALM_CPU", None)
    r = subprocess.run([oracle.RUN_HIP, os.path.join(GOLDEN, "tiny_fp16.calm"), "-x", os.path.join(GOLDEN, "sample.txt"), "-n", "48"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [l for l in r.stdout.splitlines() if l.startswith("# perplexity:")][0]
    want = open(os.path.join(GOLDEN, "cli_tiny_fp16_perplexity.txt")).read()
    g = [float(x) for x in re.findall(r"[0-9]+\.[0-9]+", got)[:2]]
    w = [float(x) for x in re.findall(r"[0-9]+\.[0-9]+", want)[:2]]
    assert abs(g[0] - w[0]) <= 5e-4 * w[0] and abs(g[1] - w[1]) <= 5e-3 * max(w[1], 1.0), (got, want)


@pytest.mark.skipif(not os.path.exists(oracle.RUN_HIP), reason="oracle/_ref/run_hip not built")
def test_perplexity_of_the_reference_text_at_mistral_width(hiplib, tmp_path):
    """SURVEY.md section 8 f3 at a BASELINE width: `run -x tools/pplx.txt` -- the reference's own perplexity text (stored here as
    the byte values the toy tokenizer maps to token ids, tests/golden/pplx_bytes.npz) -- over a 4-layer Mistral-7B-width fp8 model
    (dim 4096, hidden 14336, 32000-token classifier; rebuilt from its seed), 35161 tokens, positions wrapping every 1024:
      (i)  through the UNMODIFIED reference CLI on the HIP backend (oracle/_ref/run_hip): its "# perplexity:" line against the line
           the reference's CPU backend printed (tests/golden/make_pplx_golden.py, 16 minutes of CPU), within 5e-4;
      (ii) the same tokens through prefill_logprobs_hip (host.perplexity: the arithmetic of study(), src/run.c:286-308, on top of
           the batched scorer): the same perplexity within 1e-3."""
    import re
    import subprocess

    from conftest import GOLDEN
    from calm_amd.host import perplexity

    want_lines = open(os.path.join(GOLDEN, "cli_mistral4_pplx_perplexity.txt")).read().splitlines()
    w = [float(x) for x in re.findall(r"[0-9]+\.[0-9]+", want_lines[1])[:2]]
    n_tokens = int(want_lines[0].split()[0])
    text = np.load(os.path.join(GOLDEN, "pplx_bytes.npz"))["bytes"].tobytes()
    txt = tmp_path / "pplx.txt"
    txt.write_bytes(text)
    model_path = str(tmp_path / "mistral4.calm")
    cf.write_synth_big(model_path, cf.SPECS["mistral-7b"], "fp8", 5, 4)  # (seed and depth of the golden: make_pplx_golden.py)
    env = dict(os.environ)
    env.pop("CALM_CPU", None)
    r = subprocess.run([oracle.RUN_HIP, model_path, "-x", str(txt), "-n", "1024"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"{n_tokens} tokens" in r.stdout  # the reference tokenizer sees the same text
    got = [l for l in r.stdout.splitlines() if l.startswith("# perplexity:")][0]
    g = [float(x) for x in re.findall(r"[0-9]+\.[0-9]+", got)[:2]]
    assert abs(g[0] - w[0]) <= 5e-4 * w[0] and abs(g[1] - w[1]) <= 5e-3 * max(w[1], 1.0), (got, want_lines[1])
    # (ii) the toy tokenizer (calmfile._toy_tokenizer) through src/tokenizer.c:203-260: a printable ASCII character other than
    # backslash and double quote is its own token id; everything else is not in the vocabulary, there are no byte-fallback
    # tokens, and no two tokens concatenate to a third: dropped
    tokens = [b for b in text if 32 <= b < 127 and b not in (0x5C, 0x22)]
    assert len(tokens) == n_tokens
    model = HostModel.from_file(model_path)
    b = HipBackend(model)
    try:
        ppl, err = perplexity(b, tokens, 1024)
    finally:
        b.close()
    print(f"perplexity of the reference text, 4-layer Mistral-7B width: reference CPU {w[0]:.3f} +- {w[1]:.3f}; reference CLI on HIP {g[0]:.3f} +- {g[1]:.3f}; "
          f"prefill_logprobs_hip {ppl:.3f} +- {err:.3f}")
    assert abs(ppl - w[0]) <= 1e-3 * w[0] and abs(err - w[1]) <= 1e-2 * max(w[1], 1.0), (ppl, err, w)


# ---------------------------------------------------------------- batched prompt ingestion ------

def _kv_floats(hiplib, b, kvbits):
    """the whole K and V cache of a backend (private device layout) as float32"""
    c = b.model.config
    nbytes = c.n_layers * c.seq_len * c.head_dim * c.n_kv_heads * (kvbits // 8)
    out = []
    for ptr in (b.t.state.key_cache, b.t.state.value_cache):
        raw = np.empty(nbytes, dtype=np.uint8)
        hiplib.download_hip(raw.ctypes.data, ptr, nbytes)
        if kvbits == 16:
            out.append(raw.view(np.float16).astype(np.float32))
        else:
            out.append((raw.astype(np.uint16) << 8).view(np.float16).astype(np.float32))  # e5m2 = top byte of fp16
    return out


@pytest.mark.parametrize("kvbits", [16, 8])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_prefill_equals_the_serial_prompt_loop(hiplib, case, kvbits):
    """prefill_hip(tokens[0:T-1]) then one decode step == the reference's logits at step T-1 (teacher-forced
    golden stream), == T-1 serial FF_UPDATE_KV_ONLY steps on the same backend; the cache rows it leaves behind
    are the serial path's.  Covers every golden shape: ragged rows, bias, clip, layernorm, parallel residual,
    MoE (serial inside the call) and the wrapping rolling buffer of sink_fp16 (serial past seq_len)."""
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    T = len(toks)
    serial = HipBackend(model, kvbits=kvbits)
    batched = HipBackend(model, kvbits=kvbits)
    # the checker: the reference's own logits for the fp16 cache (golden), the oracle's fp8-KV mode for the e5m2 cache (the
    # reference's CPU path has none, src/infer.c:161; oracle/calm_oracle.c: kv_store restates the CUDA path's storage)
    o8 = oracle.OracleBackend(model, kvbits=8) if kvbits == 8 else None
    try:
        for pos, tok in enumerate(toks[: T - 1]):
            serial.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        ls = serial.forward(toks[T - 1], T - 1, 0).copy()
        batched.prefill(toks[: T - 1], 0)
        lb = batched.forward(toks[T - 1], T - 1, 0).copy()
        if kvbits == 16:
            want = z["logits"][T - 1]
            assert rel_err(lb, ls) < 2e-4, rel_err(lb, ls)
        else:
            for pos, tok in enumerate(toks[: T - 1]):
                o8.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
            want = o8.forward(toks[T - 1], T - 1, 0).copy()
        assert rel_err(lb, want) < logit_tol(kvbits), rel_err(lb, want)
        assert rel_err(ls, want) < logit_tol(kvbits), rel_err(ls, want)
        for ks, kb in zip(_kv_floats(hiplib, serial, kvbits), _kv_floats(hiplib, batched, kvbits)):
            # the same values up to one rounding step of the cache format (fp32 sums differ in their last bits)
            assert np.abs(ks - kb).max() <= (2e-3 if kvbits == 16 else 0.26) * max(np.abs(ks).max(), 1e-6)
    finally:
        serial.close()
        batched.close()
        if o8 is not None:
            o8.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_prefill_big_gemm_form_on_every_golden_shape(hiplib, case):
    """calm_hip_configure("pf_forms", 2): the FFN-up and the classifier of every dense fp8 / gf4 chunk through k_pf_gemm_big (512 units x
    128 tokens per workgroup) whatever the grid -- ragged hidden sizes, partial 128-token columns, gelu, parallel residual -- against
    the reference's golden logits and against the same prompt with the form off; scored log-probabilities of both agree."""
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    T = len(toks)
    big = HipBackend(model)
    plain = HipBackend(model)
    try:
        assert hiplib.calm_hip_configure(b"pf_forms", 2) == 0  # 2: the big form always
        try:
            big.prefill(toks[: T - 1], 0)
            lb = big.forward(toks[T - 1], T - 1, 0).copy()
            lpb = big.prefill_logprobs(toks, 0)
        finally:
            hiplib.calm_hip_configure(b"pf_forms", 1)  # 1: K-split GEMMs only
        try:
            plain.prefill(toks[: T - 1], 0)
            lp = plain.forward(toks[T - 1], T - 1, 0).copy()
            lpp = plain.prefill_logprobs(toks, 0)
        finally:
            hiplib.calm_hip_configure(b"pf_forms", 0)
        assert rel_err(lb, z["logits"][T - 1]) < LOGIT_TOL, rel_err(lb, z["logits"][T - 1])
        assert rel_err(lb, lp) < 2e-5, rel_err(lb, lp)
        assert np.isfinite(lpb).all() and np.abs(lpb - lpp).max() < 1e-4 * max(1.0, float(np.abs(lpp).max()))
    finally:
        big.close()
        plain.close()


def test_prefill_falls_back_to_the_serial_path_when_an_activation_leaves_binary16(hiplib):
    """The prompt GEMMs carry fp32 activations as hi + lo binary16 (range +-65504).  A model whose FFN hidden values exceed that
    (weights 100 x the usual scale: act(w1 x) * (w3 x) ~ 1e5..1e6) must not silently saturate: the chunk raises the range flag
    and is redone by the serial fp32 decode path, so cache rows and logits are THE SERIAL PATH'S, bit for bit -- and the
    oracle's within the usual tolerance."""
    spec = cf.tiny_spec("pfrange", max_seq_len=96, dim=128, hidden_dim=352, n_heads=4, n_kv_heads=2, head_dim=32, vocab_size=300)
    tensors, md = cf.synth_model(spec, "fp16", seed=5, sigma=12.0)
    model = HostModel(tensors, md)
    rng = np.random.default_rng(8)
    toks = [int(t) for t in rng.integers(0, 300, size=41)]
    o = oracle.OracleBackend(model)
    serial = HipBackend(model)
    batched = HipBackend(model)
    try:
        biggest = 0.0
        for pos, tok in enumerate(toks[:-1]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
            biggest = max(biggest, float(np.abs(o.state("hb", spec.hidden_dim)).max()))  # (the last layer's FFN hidden values)
            serial.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[-1], 40, 0).copy()
        assert biggest > 65504.0, "the fixture no longer leaves the binary16 range"
        ls = serial.forward(toks[-1], 40, 0).copy()
        before = hiplib.calm_hip_query(b"pf_redone", 0)
        batched.prefill(toks[:-1], 0)
        assert hiplib.calm_hip_query(b"pf_redone", 0) == before + 40, "the chunk was not sent back through the serial path"
        lb = batched.forward(toks[-1], 40, 0).copy()
        assert np.array_equal(lb, ls)
        assert rel_err(lb, lo) < LOGIT_TOL, rel_err(lb, lo)
        lp = batched.prefill_logprobs(toks, 0)  # scoring takes the same way out
        assert np.isfinite(lp).all()
        # ... and a model of the usual scale does not take it
        before = hiplib.calm_hip_query(b"pf_redone", 0)
    finally:
        o.close()
        serial.close()
        batched.close()
    model2, z = load_golden("tiny_fp8")
    b2 = HipBackend(model2)
    try:
        b2.prefill([int(t) for t in z["tokens"]][:-1], 0)
        assert hiplib.calm_hip_query(b"pf_redone", 0) == before
    finally:
        b2.close()


def test_prefill_redoes_only_from_the_first_chunk_that_left_binary16(hiplib):
    """Range flags are per chunk (advisor, round 4: one flag per call sent a whole 30k-token prompt back through the serial path for an
    overflow in its last chunk).  One layer whose attention output projection is zero, so a position's FFN hidden values depend on its
    token alone: 1024 'cool' tokens fill the first chunk, the second holds three whose hidden values exceed 65504 -- checked on the
    oracle -- and prefill_hip must redo tokens [1024, n) only, ending with the serial path's cache rows and logits."""
    spec = cf.tiny_spec("pfrange2", max_seq_len=1536, dim=128, hidden_dim=352, n_heads=4, n_kv_heads=2, head_dim=32, vocab_size=300, n_layers=1)
    tensors, md = cf.synth_model(spec, "fp16", seed=5, sigma=8.0)
    tensors["model.layers.0.attn.wo.weight"] = np.zeros_like(tensors["model.layers.0.attn.wo.weight"])
    model = HostModel(tensors, md)
    o = oracle.OracleBackend(model)
    serial = HipBackend(model)
    old_chunk = hiplib.calm_hip_configure(b"pf_chunk", 1024)  # (read when a model's prompt buffers are allocated)
    batched = HipBackend(model)
    try:
        mags = []
        for tok in range(300):
            o.forward(tok, 0, abi.FF_UPDATE_KV_ONLY)
            mags.append(float(np.abs(o.state("hb", spec.hidden_dim)).max()))
        order = np.argsort(mags)
        cool, hot = order[:60], int(order[-1])
        assert mags[int(cool[-1])] < 0.6 * 65504 and mags[hot] > 1.05 * 65504, "the fixture no longer straddles the binary16 range"
        rng = np.random.default_rng(8)
        toks = [int(t) for t in rng.choice(cool, size=1101)]
        for p_ in (1030, 1050, 1077):
            toks[p_] = hot
        for pos, tok in enumerate(toks[:-1]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
            serial.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[-1], 1100, 0).copy()
        ls = serial.forward(toks[-1], 1100, 0).copy()
        before = hiplib.calm_hip_query(b"pf_redone", 0)
        batched.prefill(toks[:-1], 0)
        assert hiplib.calm_hip_query(b"pf_redone", 0) == before + (1100 - 1024), "tokens redone: the second chunk's, no more, no fewer"
        lb = batched.forward(toks[-1], 1100, 0).copy()
        assert rel_err(lb, ls) < 2e-5, rel_err(lb, ls)  # (first-chunk rows from the batched path: fp32-rounding apart from the serial ones)
        assert rel_err(lb, lo) < LOGIT_TOL, rel_err(lb, lo)
        # the redone rows are the serial path's own, bit for bit; the first chunk's came from the matrix cores
        kb, ks = batched.read_kv(0, 1), serial.read_kv(0, 1)
        assert np.array_equal(kb[1024:1100].view(np.uint16), ks[1024:1100].view(np.uint16))
    finally:
        hiplib.calm_hip_configure(b"pf_chunk", old_chunk)
        o.close()
        serial.close()
        batched.close()


def test_scoring_in_blocks_of_a_bounded_logits_scratch(hiplib):
    """prefill_logprobs_hip scores a chunk in blocks of as many tokens as the logits scratch holds ("pf_score_mb"; 2048 tokens at a 32k
    vocabulary, 512 at 128k, 256 at 256k -- round 4 sized the scratch by the chunk: 2.1 GB at a 256k vocabulary).  A 1 MiB scratch on
    a 500-token vocabulary is 512-token blocks: a 1300-token prompt is scored in 512 + 512 + 276, same log-probabilities as in one."""
    spec = cf.tiny_spec("pfscore", max_seq_len=1536, dim=128, hidden_dim=352, n_heads=4, n_kv_heads=2, head_dim=32, vocab_size=500)
    tensors, md = cf.synth_model(spec, "fp8", seed=22)
    model = HostModel(tensors, md)
    rng = np.random.default_rng(5)
    toks = [int(t) for t in rng.integers(0, 500, size=1300)]
    whole = HipBackend(model)
    try:
        lp_whole = whole.prefill_logprobs(toks, 0).copy()
    finally:
        whole.close()
    old = hiplib.calm_hip_configure(b"pf_score_mb", 1)
    blocks = HipBackend(model)
    try:
        lp_blocks = blocks.prefill_logprobs(toks, 0).copy()
    finally:
        blocks.close()
        hiplib.calm_hip_configure(b"pf_score_mb", old)
    assert np.isfinite(lp_whole).all() and lp_whole[:-1].max() < 0
    # (the classifier GEMM's form -- hence its summation order -- follows the block's token count: fp32 rounding apart, not bit-equal)
    assert np.abs(lp_blocks - lp_whole).max() < 1e-4 * max(1.0, float(np.abs(lp_whole).max()))
    # ... and against the oracle at a few positions
    o = oracle.OracleBackend(model)
    try:
        for pos in range(1100):
            lg = o.forward(toks[pos], pos, 0 if pos in (0, 511, 512, 1023, 1024, 1099) else abi.FF_UPDATE_KV_ONLY)
            if lg is not None:
                z = lg.astype(np.float64)
                ref = z[toks[pos + 1]] - z.max() - np.log(np.exp(z - z.max()).sum())
                assert abs(lp_blocks[pos] - ref) < 2e-3, (pos, lp_blocks[pos], ref)
    finally:
        o.close()


def test_prefill_in_two_calls_and_odd_chunks(hiplib):
    """a prompt longer than one 1024-token chunk, split at awkward places, with the second call starting at pos > 0"""
    spec = cf.tiny_spec("pf", max_seq_len=1536, dim=128, hidden_dim=352, n_heads=4, n_kv_heads=2, head_dim=32, vocab_size=500)
    tensors, md = cf.synth_model(spec, "fp8", seed=21)
    model = HostModel(tensors, md)
    rng = np.random.default_rng(4)
    toks = [int(t) for t in rng.integers(0, 500, size=1300)]
    o = oracle.OracleBackend(model)
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(toks[:-1]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[-1], len(toks) - 1, 0).copy()
        b.prefill(toks[:1065], 0)         # chunks of 1024 + 41
        b.prefill(toks[1065:1299], 1065)  # one chunk of 234 (3 x 64 + 42) that attends to the 1065 rows before it
        lb = b.forward(toks[-1], len(toks) - 1, 0)
        assert rel_err(lb, lo) < LOGIT_TOL, rel_err(lb, lo)
        b.prefill([], 0)  # empty prompt: no-op
    finally:
        b.close()
        o.close()


@pytest.mark.parametrize("dtype", ["fp8", "fp16", "gf4"])
@pytest.mark.parametrize("kvbits", [16, 8])
def test_prefill_chunks_of_three_or_four_tokens_stream_the_weights_once(hiplib, dtype, kvbits):
    """chunks of 3 or 4 tokens go through k_pf_skinny (one weight stream, four tokens' activations behind it; the FFN-down in column
    ranges where its image does not fit the LDS), larger ones through the GEMM forms: a 60-token prompt fed 3, 4, 5, 6, 7, 8, 8, 4, 3,
    7, 5 tokens at a time, then a decode step against the oracle (same cache format); the same prompt with the knob off: cache rows agree"""
    spec = cf.tiny_spec("sk", max_seq_len=128, dim=1024, hidden_dim=12288, n_heads=8, n_kv_heads=2, head_dim=128, vocab_size=400, n_layers=2)
    tensors, md = cf.synth_model(spec, dtype, seed=77)
    model = HostModel(tensors, md)
    rng = np.random.default_rng(8)
    sizes = [3, 4, 5, 6, 7, 8, 8, 4, 3, 7, 5]
    toks = [int(t) for t in rng.integers(0, 400, size=sum(sizes) + 1)]
    o = oracle.OracleBackend(model, kvbits=kvbits)
    b = HipBackend(model, kvbits=kvbits)
    g = HipBackend(model, kvbits=kvbits)
    try:
        for pos, tok in enumerate(toks[:-1]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[-1], len(toks) - 1, 0).copy()
        assert hiplib.calm_hip_configure(b"pf_forms", -1) == 0
        pos = 0
        for n in sizes:
            b.prefill(toks[pos : pos + n], pos)
            pos += n
        lb = b.forward(toks[-1], pos, 0).copy()
        assert rel_err(lb, lo) < logit_tol(kvbits), rel_err(lb, lo)
        hiplib.calm_hip_configure(b"pf_forms", 32)  # no skinny chain: the GEMM forms
        try:
            pos = 0
            for n in sizes:
                g.prefill(toks[pos : pos + n], pos)
                pos += n
        finally:
            hiplib.calm_hip_configure(b"pf_forms", 0)
        lg = g.forward(toks[-1], pos, 0).copy()
        assert rel_err(lg, lo) < logit_tol(kvbits), rel_err(lg, lo)
        if kvbits == 16:
            assert rel_err(lb, lg) < 2e-5, rel_err(lb, lg)
        for ks, kg in zip(_kv_floats(hiplib, b, kvbits), _kv_floats(hiplib, g, kvbits)):
            assert np.abs(ks - kg).max() <= (2e-3 if kvbits == 16 else 0.26) * max(np.abs(kg).max(), 1e-6)
    finally:
        b.close()
        g.close()
        o.close()


@pytest.mark.parametrize("kvbits", [16, 8])
@pytest.mark.parametrize("head_dim,n_heads,n_kv_heads", [(64, 4, 4), (64, 4, 2), (64, 8, 2), (64, 12, 2), (64, 16, 2), (128, 4, 1), (128, 6, 2), (128, 2, 2)])
def test_prefill_attention_on_the_matrix_cores(hiplib, head_dim, n_heads, n_kv_heads, kvbits):
    """k_pf_attn_mfma (head sizes 64 / 128): every grouping of query heads per kv head it is compiled for -- 1, 2, 4 heads per
    round, one or several rounds, several token tiles per workgroup -- with both cache formats: a 333-token prompt in four
    calls (201, 3, 33 and 96 tokens: partial tiles on both sides, a nearly empty query tile), then one decode step against the oracle;
    and the same prompt through the lane-arithmetic kernel (calm_hip_configure("pf_forms", 16)): cache rows equal"""
    dim = 256
    # (head size 128 with a window of whole 64-position blocks: V is read from the transposed cache -- the other window length keeps the
    # LDS-transposing form of the same kernel under test)
    seq_len = 448 if (head_dim == 128 and n_heads != 6) else 400
    spec = cf.tiny_spec("pfa", max_seq_len=seq_len, dim=dim, hidden_dim=512, n_heads=n_heads, n_kv_heads=n_kv_heads, head_dim=head_dim, vocab_size=300, n_layers=2)
    tensors, md = cf.synth_model(spec, "fp8", seed=31)
    model = HostModel(tensors, md)
    rng = np.random.default_rng(6)
    toks = [int(t) for t in rng.integers(0, 300, size=334)]
    o = oracle.OracleBackend(model, kvbits=kvbits)
    b = HipBackend(model, kvbits=kvbits)
    v = HipBackend(model, kvbits=kvbits)
    try:
        for pos, tok in enumerate(toks[:-1]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[-1], 333, 0).copy()
        assert hiplib.calm_hip_configure(b"pf_forms", -1) == 0
        b.prefill(toks[:201], 0)
        b.prefill(toks[201:204], 201)  # a chunk of three tokens (the smallest that is batched: one or two go through the decode path) ...
        b.prefill(toks[204:237], 204)  # ... of one tile and one token ...
        b.prefill(toks[237:333], 237)
        lb = b.forward(toks[-1], 333, 0).copy()
        assert rel_err(lb, lo) < logit_tol(kvbits), rel_err(lb, lo)
        hiplib.calm_hip_configure(b"pf_forms", 16)
        try:
            v.prefill(toks[:201], 0)
            v.prefill(toks[201:333], 201)
        finally:
            hiplib.calm_hip_configure(b"pf_forms", 0)
        lv = v.forward(toks[-1], 333, 0).copy()
        assert rel_err(lv, lo) < logit_tol(kvbits), rel_err(lv, lo)  # (lo: the oracle with THIS cache format, fp8-KV mode included)
        if kvbits == 16:
            assert rel_err(lb, lv) < 2e-5, rel_err(lb, lv)
        for km, kv in zip(_kv_floats(hiplib, b, kvbits), _kv_floats(hiplib, v, kvbits)):
            assert np.abs(km - kv).max() <= (2e-3 if kvbits == 16 else 0.26) * max(np.abs(kv).max(), 1e-6)
    finally:
        b.close()
        v.close()
        o.close()


@pytest.mark.parametrize("name,dtype", [("mistral-7b", "fp8"), ("llama-3-8b", "gf4"), ("tinyllama-1.1b", "fp16"), ("mixtral-8x7b", "fp8")])
def test_prefill_long_prompt_takes_the_wide_gemm_form(hiplib, name, dtype):
    """BASELINE widths, one layer, a 1100-token prompt (one chunk; the mixture of experts' grouped GEMMs in the big form too): every GEMM
    of the long chunk runs in the wide form (k_pf_gemm_wide: B staged through LDS, no K split) -- the FFN-up and the classifier of the dense fp8 / gf4 models in
    the big form (k_pf_gemm_big: 512 units x 128 tokens per workgroup), as is the FFN-down with K cut into ranges -- a short chunk in the
    K-split form.  Against serial
    ingestion on the same backend, against the K-split form alone (calm_hip_configure("pf_forms", 1)), and the
    scored log-probabilities of both against each other."""
    spec = cf.SPECS[name]
    tensors, md = cf.synth_model_big(spec, dtype, seed=10, n_layers=1)
    model = HostModel(tensors, md, context=1280)
    rng = np.random.default_rng(12)
    n = 1100
    toks = [int(t) for t in rng.integers(0, spec.vocab_size, size=n + 1)]
    serial = HipBackend(model)
    wide = HipBackend(model)
    ksplit = HipBackend(model)
    try:
        for pos, tok in enumerate(toks[:n]):
            serial.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        ls = serial.forward(toks[n], n, 0).copy()
        assert hiplib.calm_hip_configure(b"pf_forms", -1) == 0
        wide.prefill(toks[:n], 0)
        lw = wide.forward(toks[n], n, 0).copy()
        lpw = wide.prefill_logprobs(toks[: n + 1], 0)
        hiplib.calm_hip_configure(b"pf_forms", 1)
        try:
            ksplit.prefill(toks[:n], 0)
            lk = ksplit.forward(toks[n], n, 0).copy()
            lpk = ksplit.prefill_logprobs(toks[: n + 1], 0)
        finally:
            hiplib.calm_hip_configure(b"pf_forms", 0)
        assert rel_err(lw, ls) < 2e-4, rel_err(lw, ls)
        assert rel_err(lk, ls) < 2e-4, rel_err(lk, ls)
        assert np.isfinite(lpw).all() and np.abs(lpw - lpk).max() < 2e-3 * max(1.0, float(np.abs(lpk).max()))
        for ks, kb in zip(_kv_floats(hiplib, serial, 16), _kv_floats(hiplib, wide, 16)):
            assert np.abs(ks - kb).max() <= 2e-3 * max(np.abs(ks).max(), 1e-6)
    finally:
        serial.close()
        wide.close()
        ksplit.close()


@pytest.mark.parametrize("dtype,experts,active,n", [("fp8", 6, 2, 3000), ("gf4", 4, 3, 3000), ("fp16", 5, 2, 3000), ("fp8", 6, 2, 4300)])
def test_prefill_mixture_chunks_of_4096_tokens_and_128_row_expert_groups(hiplib, dtype, experts, active, n):
    """Round 5: a mixture-of-experts chunk holds up to 4096 tokens (k_pf_route: four tokens per thread, packed in token order) and, for
    fp8 / gf4 weights, pads every expert's rows to whole 128-row columns so that the grouped GEMMs run in the big form (k_pf_gemm_big,
    FFN-down in ranges of K).  A 3000-token prompt as ONE chunk in that form, against the same prompt (a) in 64-row groups through the
    wide / K-split forms ("pf_forms" 4), (b) in chunks of 1024 + 1024 + 952 ("pf_chunk_moe" 1024): fp32-rounding apart; the scored
    log-probabilities likewise; and the logits behind the prompt against the oracle.  (fp16 weights keep 64-row groups: same checks.)
    The 4300-token prompt is one FULL 4096-token chunk (k_pf_route's fourth pass, every scratch buffer at its bound) and a remainder."""
    spec = cf.tiny_spec("pfmoe", max_seq_len=3072 if n < 3072 else 4352, dim=128, hidden_dim=352, n_heads=4, n_kv_heads=2, head_dim=32, vocab_size=400,
                        n_layers=2, n_experts=experts, n_experts_active=active)
    tensors, md = cf.synth_model(spec, dtype, seed=41)
    model = HostModel(tensors, md)
    rng = np.random.default_rng(14)
    toks = [int(t) for t in rng.integers(0, 400, size=n + 1)]

    def run(knobs):
        old = {k: hiplib.calm_hip_configure(k, v) for k, v in knobs.items()}
        b = HipBackend(model)
        try:
            lp = b.prefill_logprobs(toks, 0).copy()  # (also fills the cache over positions 0 .. n)
            b.prefill(toks[:n], 0)
            return b.forward(toks[n], n, 0).copy(), lp
        finally:
            b.close()
            for k, v in old.items():
                hiplib.calm_hip_configure(k, v)

    l_big, lp_big = run({})
    l_small, lp_small = run({b"pf_forms": 4})
    l_two, lp_two = run({b"pf_chunk_moe": 1024})
    assert np.isfinite(l_big).all() and np.isfinite(lp_big).all()
    assert rel_err(l_big, l_small) < 2e-4, rel_err(l_big, l_small)
    assert rel_err(l_big, l_two) < 2e-4, rel_err(l_big, l_two)
    assert np.abs(lp_big - lp_small).max() < 2e-3 * max(1.0, float(np.abs(lp_small).max()))
    assert np.abs(lp_big - lp_two).max() < 2e-3 * max(1.0, float(np.abs(lp_two).max()))
    o = oracle.OracleBackend(model)
    try:
        for pos, tok in enumerate(toks[:n]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[n], n, 0).copy()
    finally:
        o.close()
    assert rel_err(l_big, lo) < LOGIT_TOL, rel_err(l_big, lo)


@pytest.mark.parametrize("name,dtype,layers", [("mistral-7b", "fp8", 2), ("llama-3-8b", "gf4", 2), ("tinyllama-1.1b", "fp16", 2), ("mixtral-8x7b", "fp8", 1), ("dbrx-132b", "fp8", 1)])
def test_prefill_full_width_matches_serial(hiplib, name, dtype, layers):
    """BASELINE shapes at full width (1-2 layers): 100-token prompt, batched vs serial ingestion on the GPU, and
    the oracle's logits after a 12-token prompt"""
    spec = cf.SPECS[name]
    tensors, md = cf.synth_model_big(spec, dtype, seed=9, n_layers=layers)
    model = HostModel(tensors, md, context=128)
    rng = np.random.default_rng(8)
    toks = [int(t) for t in rng.integers(0, spec.vocab_size, size=101)]
    o = oracle.OracleBackend(model)
    serial = HipBackend(model)
    batched = HipBackend(model)
    try:
        for pos, tok in enumerate(toks[:12]):
            o.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        lo = o.forward(toks[12], 12, 0).copy()
        batched.prefill(toks[:12], 0)
        assert rel_err(batched.forward(toks[12], 12, 0), lo) < LOGIT_TOL
        for pos, tok in enumerate(toks[:100]):
            serial.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)
        ls = serial.forward(toks[100], 100, 0).copy()
        batched.prefill(toks[:100], 0)
        lb = batched.forward(toks[100], 100, 0)
        assert rel_err(lb, ls) < 2e-4, rel_err(lb, ls)
        # scoring at the full vocabulary (the classifier as a [vocab x dim] GEMM over the chunk): the log-probability
        # of the token that follows position 100, from the batched call, against the serial path's logits
        lp = batched.prefill_logprobs(toks[:101], 0)
        serial99 = serial.forward(toks[99], 99, 0).astype(np.float64)  # position 99's logits score toks[100]
        want = (serial99[toks[100]] - serial99.max()) - np.log(np.exp(serial99 - serial99.max()).sum())
        assert abs(float(lp[99]) - want) < 2e-3 * max(1.0, abs(want)), (float(lp[99]), want)
    finally:
        serial.close()
        batched.close()
        o.close()


@pytest.mark.parametrize("case", ["tiny_fp8", "moe_fp8", "bias_tied_gf4"])
def test_generate_with_a_batched_prompt_continues_the_reference_stream(hiplib, case):
    """the host loop with the prompt handed to prefill_hip: a 9-token prompt taken from the reference's greedy
    stream must be continued by exactly the tokens the reference produced next"""
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    b = HipBackend(model)
    try:
        out, _ = generate(b, model, toks[:9], len(toks), batched_prompt=True)
        assert out[:-1] == toks[1:]
    finally:
        b.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 33])
def test_prefill_of_very_short_and_odd_prompts(hiplib, n):
    """1, 2 tokens (taken through the decode path: a batched chunk costs about three decode steps), 3, 5 (partial query groups
    of the attention kernel) and 33 (one token past an MFMA column tile),
    each followed by one decode step checked against the reference's logits; the 33-token case wraps the
    16-slot rolling buffer of sink_fp16 inside the call"""
    case = "sink_fp16" if n == 33 else "ragged_fp8"
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    if n >= len(toks):
        pytest.skip("golden stream too short")
    b = HipBackend(model)
    try:
        b.prefill(toks[:n], 0)
        assert rel_err(b.forward(toks[n], n, 0), z["logits"][n]) < LOGIT_TOL
    finally:
        b.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_prefill_logprobs_match_the_reference_logits(hiplib, case):
    """prefill_logprobs_hip: log softmax(logits_i)[token_{i+1}] for the teacher-forced golden stream, against the same
    quantity computed from the REFERENCE's logits (src/run.c:294-298 with sample_prob of src/sampler.c:19-32);
    the sink cases score their positions past seq_len through the serial path inside the call (rolling buffer wrapped)"""
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    ref = z["logits"].astype(np.float64)
    want = np.array([(ref[i] - ref[i].max())[toks[i + 1]] - np.log(np.exp(ref[i] - ref[i].max()).sum()) for i in range(len(toks) - 1)])
    b = HipBackend(model)
    try:
        got = b.prefill_logprobs(toks, 0)
        assert got.shape == want.shape
        tol = 2 * LOGIT_TOL * np.abs(ref).max()
        assert np.abs(got - want).max() < tol, (np.abs(got - want).max(), tol)
        if len(toks) <= model.config.seq_len:  # (past seq_len a repeated position would rotate the sink keys a second time)
            lg = b.forward(toks[-1], len(toks) - 1, 0)  # the cache it leaves behind is the serial loop's
            assert rel_err(lg, z["logits"][-1]) < LOGIT_TOL
        ppl = float(np.exp(-got.astype(np.float64).mean()))
        assert abs(ppl - float(np.exp(-want.mean()))) < 1e-2 * float(np.exp(-want.mean()))
    finally:
        b.close()


def test_perplexity_windows_on_the_gpu_equal_the_cpu_loop(hiplib):
    """host.perplexity (the reference's study() arithmetic, src/run.c:286-308) with every window scored by one
    prefill_logprobs_hip call, against the same windows walked one forward() at a time by the CPU oracle"""
    from calm_amd.host import perplexity

    model, z = load_golden("tiny_fp8")
    rng = np.random.default_rng(11)
    toks = [int(t) for t in rng.integers(0, model.config.vocab_size, size=61)]
    o = oracle.OracleBackend(model)
    b = HipBackend(model)
    try:
        want, want_err = perplexity(o, toks, 16)
        got, got_err = perplexity(b, toks, 16)
        assert abs(got - want) < 2e-3 * want and abs(got_err - want_err) < 2e-2 * max(want_err, 1e-6)
    finally:
        b.close()
        o.close()
