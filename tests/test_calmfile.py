""".calm container, quantisers and synthetic models (host side, no GPU)."""
import os
import subprocess

import numpy as np
import pytest

from calm_amd import calmfile as cf
from calm_amd.host import HostModel
from conftest import GOLDEN, REFERENCE


def test_fp8_matches_torch_cast():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.standard_normal(50000).astype(np.float32) * s for s in (1e-6, 1e-3, 1, 300, 60000)])
    a = np.concatenate([a, np.array([0, -0.0, np.inf, -np.inf, 65504, 57344, 61440, 61439.9, 1e10, 2**-16, 2**-17, 1.5 * 2**-17, 2**-18], dtype=np.float32)])
    ours = cf.f32_to_fp8_e5m2(a)
    ref = torch.from_numpy(a).to(torch.float8_e5m2).view(torch.uint8).numpy()
    assert np.array_equal(ours, ref)
    assert np.array_equal(cf.fp8_e5m2_to_f32(ref), torch.from_numpy(ref).view(torch.float8_e5m2).float().numpy(), equal_nan=True)


def test_fp8_is_top_byte_of_half():
    b = np.arange(256, dtype=np.uint8)
    f = cf.fp8_e5m2_to_f32(b)
    h = (b.astype(np.uint16) << 8).view(np.float16)
    assert np.array_equal(f, h.astype(np.float32), equal_nan=True)


def test_gf4_round_trip_properties():
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((8, 256)) * 0.05).astype(np.float16).astype(np.float32)
    w[0, :8] = 0  # an all-zero group
    words = cf.quantize_gf4(w)
    assert words.dtype == np.int32 and words.shape == (8, 32)
    d = cf.gf4_to_f32(words)
    g, dg = w.reshape(8, 32, 8), d.reshape(8, 32, 8)
    # the max-magnitude element of each group decodes to the e5m2-rounded max (code 0)
    idx = np.abs(g).argmax(-1)
    gmax = np.take_along_axis(g, idx[..., None], -1)[..., 0]
    dmax = np.take_along_axis(dg, idx[..., None], -1)[..., 0]
    assert np.array_equal(dmax, cf.fp8_e5m2_to_f32(cf.f32_to_fp8_e5m2(gmax)))
    # everything else: half a step (|S|/8) in range; the worst case is an element of opposite sign and
    # near-max magnitude, clamped to code 7 = -0.75 S while S was rounded down by 1/8: 0.393 |S|
    S = np.abs(dmax)[..., None]
    assert (np.abs(dg - g) <= 0.41 * S + 1e-12).all()
    same = np.sign(g) == np.sign(np.take_along_axis(g, idx[..., None], -1))
    assert (np.abs(dg - g)[same] <= (S / 8 * 1.001 + 1e-12 + 0 * g)[same]).all()
    assert np.all(d[0, :8] == 0)


def test_write_read_round_trip(tmp_path):
    spec = cf.tiny_spec(n_experts=4, n_experts_active=2)
    tensors, md = cf.synth_model(spec, "fp8", seed=7)
    p = str(tmp_path / "m.calm")
    cf.write_calm(p, tensors, md)
    f = cf.CalmFile(p)
    assert f.metadata["dtype"] == "fp8" and int(f.metadata["n_experts"]) == 4
    assert f.dtype_tag("model.embed.weight") == "F8_E5M2"
    assert f.dtype_tag("model.layers.0.attn.norm.weight") == "F32"
    for n, a in tensors.items():
        assert np.array_equal(f.tensor(n).view(np.uint8).reshape(-1), np.ascontiguousarray(a).view(np.uint8).reshape(-1)), n
    m = HostModel.from_file(p)
    assert m.config.n_experts == 4 and m.config.n_experts_ac == 2 and m.config.seq_len == 64
    assert m.tensors["model.layers.1.mlp.w1.weight"].shape == (4, 160, 64)
    f.close()


def test_data_area_is_256_aligned(tmp_path):
    p = str(tmp_path / "m.calm")
    cf.write_synth(p, cf.tiny_spec(), "gf4", seed=1)
    import struct

    (hsize,) = struct.unpack("<Q", open(p, "rb").read(8))
    assert (8 + hsize) % 256 == 0


def test_golden_models_are_reproducible():
    """the committed fixtures come from this very generator (seed 1234)"""
    import hashlib
    import tempfile

    for name, kw, dtype in [("tiny_fp8", {}, "fp8"), ("bias_tied_gf4", dict(qkv_bias=True, tied=True), "gf4")]:
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "x.calm")
            cf.write_synth(p, cf.tiny_spec(name, **kw), dtype, seed=1234)
            assert hashlib.sha256(open(p, "rb").read()).hexdigest() == hashlib.sha256(open(os.path.join(GOLDEN, name + ".calm"), "rb").read()).hexdigest()


def test_accounting_matches_reference_formula():
    """n_bandwidth of src/run.c:523-532 on the BASELINE shapes, from shapes alone"""
    nb = cf.spec_accounting(cf.SPECS["mistral-7b"], "fp8")["n_bandwidth"]
    assert nb == 32 * 218_103_808 + 131_072_000 + 1_064_960 == 7_111_458_816
    mix = cf.spec_accounting(cf.SPECS["mixtral-8x7b"], "fp8")
    assert abs(mix["n_bandwidth"] / 1e9 - 12.750) < 0.01
    tiny = cf.tiny_spec(n_experts=4, n_experts_active=2)
    t, md = cf.synth_model(tiny, "gf4", seed=0)
    m = HostModel(t, md)
    acc = cf.spec_accounting(tiny, "gf4")
    assert (acc["n_params"], acc["n_bytes"], acc["n_bandwidth"]) == m.accounting()


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "run_cpu")), reason="reference CLI not built on this box")
def test_reference_cli_reads_our_files_and_reproduces_golden_text():
    """the reference's own parser/tokenizer/sampler/CPU backend accept a file written by write_calm"""
    from oracle import oracle

    env = dict(os.environ, CALM_CPU="1", OMP_NUM_THREADS="2")
    r = subprocess.run([oracle.RUN_CPU, os.path.join(GOLDEN, "tiny_fp16.calm"), "-i", "abc abc", "-t", "0", "-n", "32"], env=env, capture_output=True, text=True, check=True)
    assert r.stdout.splitlines()[1] + "\n" == open(os.path.join(GOLDEN, "cli_tiny_fp16.txt")).read()


@pytest.mark.parametrize("dtype", ["fp16", "fp8", "gf4"])
def test_streamed_writer_equals_the_in_memory_model(tmp_path, dtype):
    """write_synth_big (one tensor in memory at a time) writes the file synth_model_big describes, tokenizer included"""
    spec = cf.ModelSpec("streamed", 128, 352, 32, 3, 4, 2, 500, 1e4, max_seq_len=64, n_experts=4 if dtype == "fp8" else 0, n_experts_active=2 if dtype == "fp8" else 0)
    path = str(tmp_path / "m.calm")
    size = cf.write_synth_big(path, spec, dtype, seed=5, n_layers=2)
    assert size == os.path.getsize(path)
    want, md = cf.synth_model_big(spec, dtype, seed=5, n_layers=2)
    f = cf.CalmFile(path)
    try:
        assert list(f.names()) == list(want)
        assert f.metadata == {k: str(v) for k, v in md.items()} and f.metadata["n_layers"] == "2"
        for name, a in want.items():
            assert f.tensor(name).tobytes() == np.ascontiguousarray(a).tobytes(), name
    finally:
        f.close()
    m = HostModel.from_file(path)
    assert m.config.n_layers == 2 and m.config.dim == 128 and m.accounting() == HostModel(want, md).accounting()


def test_streamed_writer_rejects_a_stream_that_leaves_the_layout(tmp_path):
    spec = cf.tiny_spec()
    layout = cf.stub_tensors(spec, "fp8")
    stream = list(cf.synth_stream_big(spec, "fp8", 1, reuse=False))
    with pytest.raises(ValueError):
        cf.write_calm_stream(str(tmp_path / "a.calm"), layout, iter(stream[:3] + stream[4:]), spec.metadata("fp8"))
    with pytest.raises(ValueError):
        cf.write_calm_stream(str(tmp_path / "b.calm"), layout, iter(stream[:5]), spec.metadata("fp8"))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "run_cpu")), reason="reference CLI not built on this box")
def test_reference_cli_decodes_a_streamed_file(tmp_path):
    """the reference CLI (its own safetensors parser, tokenizer and CPU backend) runs a file written by the streaming
    writer, and produces the token stream our oracle produces for the same model"""
    from oracle import oracle

    spec = cf.ModelSpec("streamed", 128, 352, 32, 2, 4, 2, 300, 1e4, max_seq_len=64)
    path = str(tmp_path / "m.calm")
    cf.write_synth_big(path, spec, "fp8", seed=2)
    env = dict(os.environ, CALM_CPU="1", OMP_NUM_THREADS="2")
    r = subprocess.run([oracle.RUN_CPU, path, "-i", "ab", "-t", "0", "-n", "12"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "tok/s" in r.stderr


@pytest.mark.skipif(not os.path.exists("/root/reference/tools/convert.py"), reason="reference converter not present on this box")
def test_quantize_gf4_is_the_reference_converters_gf4():
    """calmfile.quantize_gf4 against the reference's own gf4() (tools/convert.py:247-268), EXECUTED from the reference tree:
    bit-equal words on random tensors of several scales and on the edge rows (clamp at code 7, zero and overflowing scales)"""
    import torch

    src = open("/root/reference/tools/convert.py").read().splitlines()
    a = next(i for i, l in enumerate(src) if l.startswith("def gf4(t):"))
    b = next(i for i in range(a, len(src)) if src[i].strip() == "return gtr.cpu()")
    ns = {"torch": torch}
    exec("\n".join(src[a : b + 1]), ns)
    rng = np.random.default_rng(11)
    parts = [rng.standard_normal((512, 64)).astype(np.float32) * s for s in (0.02, 1.0, 300.0, 1e-6)]
    edge = np.zeros((8, 64), dtype=np.float32)
    edge[1, :8] = [1, -1, 0.5, -0.5, 0.25, -0.25, 0.874, -0.876]
    edge[2, :8] = [-2, 2, 1, -1, 0.24, 0.26, 1.74, 1.76]
    edge[3, :8] = 1e-30
    edge[4, :8] = [70000, 1, 2, 3, 4, 5, 6, 7]
    w = np.concatenate(parts + [edge])
    want = ns["gf4"](torch.from_numpy(w.copy())).numpy().astype(np.int32)
    got = cf.quantize_gf4(w)
    assert got.shape == want.shape
    bad = np.nonzero(got != want)
    assert bad[0].size == 0, (bad[0][:5], bad[1][:5])


def test_outlier_fixture_has_the_statistics_it_claims():
    """synth_model_big(outliers=True): a handful of residual channels 1-2 orders of magnitude above the rest by mid-depth, heavy-tailed
    weights, log-normal norm weights -- measured with the CPU oracle on the TinyLlama shape at half depth (the GPU tests run the
    BASELINE shapes at full depth on it, tests/test_full_depth_parity.py)"""
    from calm_amd.host import HostModel
    from oracle import oracle

    spec = cf.SPECS["tinyllama-1.1b"]
    tensors, md = cf.synth_model_big(spec, "fp8", seed=5, n_layers=11, outliers=True)
    plain, _ = cf.synth_model_big(spec, "fp8", seed=5, n_layers=1)
    w = cf.dequantize(tensors["model.layers.3.mlp.w1.weight"], "fp8").ravel()
    w0 = cf.dequantize(plain["model.layers.0.mlp.w1.weight"], "fp8").ravel()
    sd = 1.0 / np.sqrt(spec.dim)
    assert (np.abs(w) > 5 * sd).mean() > 20 * max((np.abs(w0) > 5 * sd).mean(), 1e-6)  # Student-t tails
    g = tensors["model.layers.3.attn.norm.weight"]
    assert g.min() > 0 and g.max() / g.min() > 5  # log-normal: an order of magnitude between channels
    ch = cf.outlier_channels(spec, 5)
    be = oracle.OracleBackend(HostModel(tensors, md, context=32))
    try:
        big, rest = [], []
        for pos in range(4):
            be.forward_stage(11 + 37 * pos, pos, 0, 1)  # first stage, not the last: state.x = the residual stream after 11 layers
            x = be.state("x", spec.dim).copy()
            big.append(np.abs(x[ch]).max())
            rest.append(np.sqrt((np.delete(x, ch) ** 2).mean()))
        assert min(big) > 60 and max(rest) < min(big) / 4, (big, rest)
    finally:
        be.close()
