"""Chained launches (CALM_HIP_CHAIN; kernels.hip.h: ChainArgs): the matvec kernels between one attention and the next
run without queue barriers between them, ordered by device-side completion counters.  The arithmetic is the ordinary
step's, so logits must be bit-identical to it.  Kept in a file of its own that sorts last: the mode is experimental.
"""
import numpy as np
import pytest

from calm_amd import abi
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel, argmax_first
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3

# dense models whose FFN has its own norm: the ones the chained-launch step accepts (mixture-of-experts and
# parallel-residual models fall back to the ordinary step, which the last two cases check)
CHAIN_CASES = ["tiny_fp16", "tiny_fp8", "tiny_gf4", "ln_gelu_clip_fp16", "bias_tied_gf4", "sink_fp16", "ragged_fp8", "partial_rope_fp16", "mqa_hd96_fp16",
               "hd256_sink_fp8", "moe_fp8", "par_fp8"]


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("case", CHAIN_CASES)
def test_chained_launches_reproduce_the_reference_logits(hiplib, case, level):
    """CALM_HIP_CHAIN: attn_out -> ffn_up -> ffn_down -> next qkv / classifier launched without queue barriers, ordered by
    device-side completion counters (kernels.hip.h: ChainArgs).  Same arithmetic, so the logits must equal the graph
    path's bit for bit, and the reference's within tolerance; KV-only steps interleaved as a prompt would."""
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    g = HipBackend(model)
    try:
        plain = [g.forward(tok, pos, 0).copy() for pos, tok in enumerate(toks)]
    finally:
        g.close()
    hiplib.calm_hip_configure(b"chain", level)  # 1: matvec kernels chained; 2: the attention kernel as well
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(toks):
            if pos in (1, 2):  # a prompt-style step: cache only, no logits, not synchronised
                assert b.forward(tok, pos, abi.FF_UPDATE_KV_ONLY) is None
                continue
            lg = b.forward(tok, pos, 0)
            assert rel_err(lg, z["logits"][pos]) < LOGIT_TOL, pos
            assert np.array_equal(lg, plain[pos]), pos
        k = b.read_kv(0, 0).astype(np.float32)
        kg = z["k_last"].view(np.float16).astype(np.float32)
        assert np.abs(k - kg).max() <= 2e-3 * max(np.abs(kg).max(), 1.0)
    finally:
        b.close()
        hiplib.calm_hip_configure(b"chain", 0)


@pytest.mark.parametrize("level", [1, 2])
def test_chained_launches_full_width(hiplib, level):
    """the chained step on whole-KiB rows at Mistral-7B width (the FULL kernels, 512 workgroups per launch)"""
    spec = cf.SPECS["mistral-7b"]
    tensors, md = cf.synth_model_big(spec, "fp8", seed=3, n_layers=3)
    model = HostModel(tensors, md, context=64)
    b = HipBackend(model)
    try:
        toks, plain = [11], []
        for pos in range(8):
            lg = b.forward(toks[-1], pos, 0)
            plain.append(lg.copy())
            toks.append(argmax_first(lg))
        b.close()
        hiplib.calm_hip_configure(b"chain", level)
        b = HipBackend(model)
        for rep in range(3):  # the counters keep counting across sequences
            for pos in range(8):
                lg = b.forward(toks[pos], pos, 0)
                assert np.array_equal(lg, plain[pos]), (rep, pos)
    finally:
        b.close()
        hiplib.calm_hip_configure(b"chain", 0)
