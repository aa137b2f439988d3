"""Host-side logic: config parsing, the generate loop and its accounting (no GPU)."""
import numpy as np

from calm_amd import abi
from calm_amd import calmfile as cf
from calm_amd.host import HostModel, argmax_first, generate, perplexity
from conftest import load_golden
from oracle import oracle


def test_config_defaults_and_caps():
    t, md = cf.synth_model(cf.tiny_spec(max_seq_len=8192), "fp16")
    m = HostModel(t, md)
    assert m.config.seq_len == 4096  # src/run.c:41-43 caps the context unless -c is given
    assert HostModel(t, md, context=512).config.seq_len == 512
    assert m.config.qkv_clip == np.finfo(np.float32).max and not m.config.norm_ln and not m.config.act_gelu
    t, md = cf.synth_model(cf.tiny_spec(norm_type="layernorm_par", act_type="gelu", qkv_clip=8.0), "fp8")
    m = HostModel(t, md)
    assert m.config.norm_ln and m.config.norm_par and m.config.act_gelu and m.config.qkv_clip == 8.0


def test_fill_transformer_tied_and_optional_tensors():
    t, md = cf.synth_model(cf.tiny_spec(tied=True, qkv_bias=True), "fp8")
    m = HostModel(t, md)
    tr = abi.Transformer()
    addrs = {n: i + 1 for i, n in enumerate(t)}
    m.fill_transformer(tr, lambda n: addrs[n])
    assert tr.weights.wcls == tr.weights.token_embedding_table == addrs["model.embed.weight"]
    assert tr.weights.bqkv[1] == addrs["model.layers.1.attn.wqkv.bias"] and not tr.weights.moegate[0]
    assert tr.weights.dbits == 8 and tr.state.kvbits == 16


def test_generate_loop_matches_reference_token_stream():
    """generate() over the oracle backend reproduces the reference's greedy stream (golden tokens)"""
    model, z = load_golden("tiny_fp8")
    o = oracle.OracleBackend(model)
    steps = len(z["tokens"])
    out, stats = generate(o, model, [int(z["tokens"][0])], steps)
    assert out[:-1] == [int(t) for t in z["tokens"][1:]]
    _, _, n_bw = model.accounting()
    assert stats["tokens"] == steps
    assert stats["read_bytes"] == steps * n_bw + sum(model.kv_bandwidth(16, p) for p in range(steps))


def test_generate_prompt_positions_are_kv_only():
    calls = []

    class Fake:
        def forward(self, token, pos, flags):
            calls.append((token, pos, flags))
            return None if flags else np.array([0.0, 2.0, 1.0], dtype=np.float32)

    model, _ = load_golden("tiny_fp8")
    out, _ = generate(Fake(), model, [7, 8, 9], 5)
    assert calls == [(7, 0, 1), (8, 1, 1), (9, 2, 0), (1, 3, 0), (1, 4, 0)]
    assert out == [8, 9, 1, 1, 1]
    assert argmax_first(np.array([1.0, 5.0, 5.0])) == 1


def test_generate_with_batched_prompt_is_the_same_loop():
    """batched_prompt=True hands the prompt positions to backend.prefill in ONE call and must yield the same
    tokens and the same byte accounting as the serial prompt loop"""
    model, z = load_golden("tiny_fp16")
    toks = [int(t) for t in z["tokens"]]
    a, b = oracle.OracleBackend(model), oracle.OracleBackend(model)
    calls = []
    real = b.prefill
    b.prefill = lambda tokens, pos: (calls.append((list(tokens), pos)), real(tokens, pos))[1]
    out_a, st_a = generate(a, model, toks[:7], len(toks))
    out_b, st_b = generate(b, model, toks[:7], len(toks), batched_prompt=True)
    assert out_a == out_b and out_a[:-1] == toks[1:]
    assert calls == [(toks[:6], 0)]
    assert st_a["read_bytes"] == st_b["read_bytes"] and st_a["tokens"] == st_b["tokens"]
    a.close(), b.close()


def test_perplexity_matches_the_reference_cli_arithmetic():
    """perplexity() over windows of `steps` tokens == the reference's study() loop: one forward per token with
    pos = i % steps (src/run.c:294-308), here both over the oracle backend"""
    import math

    model, z = load_golden("tiny_fp16")
    rng = np.random.default_rng(3)
    toks = [int(t) for t in rng.integers(0, model.config.vocab_size, size=37)]
    steps = 10
    o = oracle.OracleBackend(model)
    s = ss = den = 0.0
    for i in range(len(toks) - 1):
        lg = o.forward(toks[i], i % steps, 0)
        e = np.exp(lg - lg.max(), dtype=np.float32)
        lp = math.log(float(e[toks[i + 1]] / e.sum(dtype=np.float32)))
        s, ss, den = s + lp, ss + lp * lp, den + 1
    want = math.exp(-s / den)
    want_err = want * math.sqrt((ss - s * s / den) / den / den)
    got, got_err = perplexity(oracle.OracleBackend(model), toks, steps)
    assert abs(got - want) < 1e-4 * want and abs(got_err - want_err) < 1e-3 * max(want_err, 1e-6)
    o.close()


def test_profile_summary_names_kernels_in_anonymous_namespaces():
    """tools/prof_summary.py: rocprofv3 prints `void (anonymous namespace)::k<...>(args)`; the summary's key is `k<...>` (round 4
    cut at the first parenthesis and filed every such kernel -- 10 % of the traced time -- under an empty name)"""
    import importlib.util
    import os

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("prof_summary", os.path.join(ROOT, "tools", "prof_summary.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    assert ps.short("void (anonymous namespace)::k_pf_gemm_big<8, 0>(PfGemmArgs)") == "k_pf_gemm_big<8, 0>"
    assert ps.short("void k_ffn_up<8, 4, true, 0, true>(float const*, float const*, int)") == "k_ffn_up<8, 4, true, 0, true>"
    assert ps.short("void (anonymous namespace)::k_vt_backfill<16>(void const*, void*) [clone .kd]") == "k_vt_backfill<16>"
    assert ps.short("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"
