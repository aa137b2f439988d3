"""The prompt GEMM (calm_amd/csrc/prefill.hip.h) on its own, through libcalm_hip_test.so: out[token][unit] = x[token] . W[unit] for a
chunk of tokens, W in each of the three weight formats, the fp32 activations carried as hi + lo binary16 on the f16 matrix cores.

Checker: the oracle's matvec (the reference's dotprod arithmetic, src/infer.c:44-140) token by token.  Both kernel forms and every
shape of the wide form's tail: ragged units (M not a multiple of the tile), ragged rows (K a multiple of 32 only), partial token
tiles, K cut into ranges across workgroups (the fold must not depend on arrival order: two runs are bit-equal).
"""
import numpy as np
import pytest

from conftest import rel_err
from calm_amd import calmfile as cf
from calm_amd.host import fptr
from oracle import oracle
from test_hip_parity import _rand_w

pytestmark = pytest.mark.gpu

GEMM_TOL = 5e-6  # max |err| / max |y|: the fp32 matvec's own rounding at these lengths; the hi + lo split adds ~1e-7

FORMS = {"ksplit-1": -1, "ksplit-2": 0, "ksplit-3": -3, "wide": 1, "wide/2": 2, "wide/5": 5, "wide/8": 8, "big": 9, "big/2": 92, "big/4": 94}


def gemm(hiplib, dtype, w, x, M, K, form):
    nb = x.shape[0]
    out = np.full((nb, M), np.nan, dtype=np.float32)
    hiplib.calm_hip_test_pf_gemm(cf.DBITS[dtype], w.ctypes.data, fptr(x), fptr(out), M, K, nb, form)
    return out


def oracle_gemm(dtype, w, x, M, K):
    return np.stack([oracle.matvec(w, x[t], cf.DBITS[dtype], K, M) for t in range(x.shape[0])])


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("dtype", ["fp16", "fp8", "gf4"])
@pytest.mark.parametrize("M,K,nb", [(256, 1024, 64), (300, 1056, 70), (1000, 4128, 37), (516, 2048, 200)])
def test_prompt_gemm_matches_the_oracle(hiplib, dtype, form, M, K, nb):
    if K % (128 // cf.DBITS[dtype]):
        pytest.skip("row not a whole number of 16-byte pieces for this format")
    if form.startswith("big") and dtype == "fp16":
        pytest.skip("the big form takes fp8 / gf4 weights (an fp16 step of A does not fit its rings)")
    rng = np.random.default_rng(M + K + nb)
    w = _rand_w(rng, M, K, dtype)
    x = rng.standard_normal((nb, K)).astype(np.float32)
    got = gemm(hiplib, dtype, w, x, M, K, FORMS[form])
    want = oracle_gemm(dtype, w, x, M, K)
    assert np.isfinite(got).all()
    assert rel_err(got, want) < GEMM_TOL, rel_err(got, want)


@pytest.mark.parametrize("form", ["wide/2", "wide/8", "big/4"])
def test_ranges_fold_in_a_fixed_order(hiplib, form):
    """K cut into ranges: whichever workgroup arrives last folds the partial tiles in range order -- two runs are bit-equal"""
    rng = np.random.default_rng(3)
    M, K, nb = 1024, 8192, 128
    w = _rand_w(rng, M, K, "fp8")
    x = rng.standard_normal((nb, K)).astype(np.float32)
    a = gemm(hiplib, "fp8", w, x, M, K, FORMS[form])
    # Repeated: the hand-off of the partial tiles is write-through (sc1) stores, a drained vmcnt and a relaxed agent-scope
    # counter, the reader's loads agent-scope too -- the CDNA guide's "drained sc1 flag" form, a contract of gfx942 / gfx950's
    # caches rather than of the memory model (prefill.hip.h: k_pf_gemm_wide).  A stale partial would show as a bit difference.
    for _ in range(24):
        assert np.array_equal(a, gemm(hiplib, "fp8", w, x, M, K, FORMS[form]))


def test_activation_split_range(hiplib):
    """x = hi + lo keeps 22 bits over the binary16 range: activations from 1e-3 to 6e4 in magnitude in one vector, and exact
    zeros; beyond +-65504 the split saturates (documented limit): the result stays finite"""
    rng = np.random.default_rng(5)
    M, K, nb = 256, 2048, 64
    w = _rand_w(rng, M, K, "fp8")
    x = (rng.standard_normal((nb, K)) * np.exp(rng.uniform(np.log(1e-3), np.log(1.5e4), size=(nb, K)))).astype(np.float32)
    x = np.clip(x, -6e4, 6e4)
    x[:, ::7] = 0.0
    got = gemm(hiplib, "fp8", w, x, M, K, FORMS["wide"])
    assert rel_err(got, oracle_gemm("fp8", w, x, M, K)) < GEMM_TOL
    x[0, 5] = 1e6
    x[1, 9] = -3e38
    sat = gemm(hiplib, "fp8", w, x, M, K, FORMS["wide"])
    assert np.isfinite(sat).all()
    assert np.array_equal(sat[2:], got[2:])


def test_full_width_shapes(hiplib):
    """the four GEMM shapes of a Mistral-7B layer cut to 512 units, 96 tokens, both forms against each other and the oracle"""
    rng = np.random.default_rng(8)
    for K in (4096, 14336):
        M, nb = 512, 96
        w = _rand_w(rng, M, K, "fp8")
        x = rng.standard_normal((nb, K)).astype(np.float32)
        want = oracle_gemm("fp8", w, x, M, K)
        for form in ("ksplit-2", "wide", "wide/5", "big", "big/2"):
            assert rel_err(gemm(hiplib, "fp8", w, x, M, K, FORMS[form]), want) < GEMM_TOL, (K, form)
