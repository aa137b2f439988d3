"""Shapes the reference's CPU path cannot run (so no golden exists): the oracle alone is the checker.  Sorts late on
purpose: new, oracle-only coverage runs after the reference-pinned tests.
"""
import numpy as np
import pytest

from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel
from conftest import LOGIT_TOL, rel_err
from oracle import oracle

pytestmark = pytest.mark.gpu



@pytest.mark.parametrize("dtype", ["fp16", "fp8", "gf4"])
def test_heads_wider_than_the_model_match_oracle(hiplib, dtype):
    """q_dim > dim > hidden_dim (Gemma-like proportions).  The reference's CPU path cannot run this shape -- its attention
    output lands in xb2 (dim floats) and wo's result in hb (hidden_dim floats), src/infer.c:152-153,404,410 -- so there is
    no golden for it; the oracle sizes those buffers by the larger dimension and is the checker here."""
    spec = cf.tiny_spec("wide_heads", dim=64, hidden_dim=32, head_dim=32, n_heads=4, n_kv_heads=2, vocab_size=200, max_seq_len=32)
    tensors, md = cf.synth_model(spec, dtype, seed=21)
    model = HostModel(tensors, md)
    o = oracle.OracleBackend(model)
    b = HipBackend(model)
    try:
        tok = 7
        for pos in range(12):
            lo = o.forward(tok, pos, 0)
            lg = b.forward(tok, pos, 0)
            assert rel_err(lg, lo) < LOGIT_TOL, (pos, rel_err(lg, lo))
            tok = oracle.argmax(lo)
        toks = [int(t) for t in np.random.default_rng(3).integers(0, 200, size=9)]
        o2, b2 = oracle.OracleBackend(model), HipBackend(model)
        o2.prefill(toks, 0)
        b2.prefill(toks, 0)
        assert rel_err(b2.forward(5, 9, 0), o2.forward(5, 9, 0)) < LOGIT_TOL
        b2.close()
        o2.close()
    finally:
        b.close()
        o.close()
