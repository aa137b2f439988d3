"""The sampler (src/sampler.c): greedy and min-p, on the host side of the checker and on the device.

* CPU: the oracle's restatement (oracle_sample) against the reference's own `sample()` (src/sampler.c compiled into
  oracle/_ref/libcalm_ref.so): the same token and the same generator state for every draw.  The reference is built
  -ffast-math, so its running sums are in whatever order its compiler chose; draws are compared exactly and the rare
  boundary case is bounded, not excused.
* GPU: k_sample_minp (libcalm_hip_test.so hook) and decode_sample_hip (the product entry point) against the oracle.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from calm_amd import abi
from oracle import oracle

CASES = [(1.0, 0.1), (0.7, 0.05), (1.5, 0.3), (0.3, 0.5), (1.0, 1e-4), (2.0, 0.9)]


def logits_like_a_model(rng, n, spread):
    """a few strong candidates over a broad floor, as classifier outputs look"""
    lg = (rng.standard_normal(n) * spread).astype(np.float32)
    hot = rng.integers(0, n, size=8)
    lg[hot] += rng.uniform(2.0, 6.0, size=8).astype(np.float32) * spread
    return lg


def numpy_minp(logits, temperature, minp, coin):
    """src/sampler.c:44-78 in float64 -> (token, distance of the draw to the nearest boundary, relative to the total)"""
    lg = logits.astype(np.float64)
    mx = lg.max()
    keep = logits >= np.float32(mx + np.float32(np.log(np.float32(minp))) * np.float32(temperature))
    p = np.where(keep, np.exp((lg - mx) / temperature), 0.0)
    cdf = np.cumsum(p)
    r = coin * cdf[-1]
    tok = int(np.searchsorted(cdf, r, side="right"))
    edges = cdf[keep]
    return min(tok, len(lg) - 1), float(np.abs(edges - r).min() / cdf[-1])


def coin_of(state):
    """src/sampler.c:7-18 -> (coin, next state)"""
    m = (1 << 64) - 1
    s = state
    s ^= s >> 12
    s ^= (s << 25) & m
    s ^= s >> 27
    u = ((s * 0x2545F4914F6CDD1D) & m) >> 32
    return np.float32(u >> 8) / np.float32(16777216.0), s


def test_oracle_greedy_cases_draw_no_coin():
    lg = np.array([0.5, 2.0, 2.0, -1.0], dtype=np.float32)
    assert oracle.sample(lg, 0.0, 0.1, 1234) == (1, 1234)  # temperature 0 (src/sampler.c:81)
    assert oracle.sample(lg, 1.0, 1.0, 1234) == (1, 1234)  # minp >= 1


def test_oracle_sample_follows_the_definition():
    rng = np.random.default_rng(3)
    state = 0x1234567
    for temperature, minp in CASES:
        for _ in range(40):
            lg = logits_like_a_model(rng, 1000, 2.0)
            coin, nxt = coin_of(state)
            want, margin = numpy_minp(lg, temperature, minp, float(coin))
            tok, state2 = oracle.sample(lg, temperature, minp, state)
            assert state2 == nxt
            assert tok == want or margin < 1e-5, (temperature, minp, tok, want, margin)
            state = state2


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built")
def test_oracle_sample_matches_the_reference_sampler():
    R = oracle.ref_lib()
    if not hasattr(R, "sample"):
        pytest.skip("oracle/_ref/libcalm_ref.so predates the sampler pin: make -C oracle ref")
    R.sample.restype = C.c_int
    R.sample.argtypes = [C.POINTER(abi.Sampler), C.POINTER(C.c_float)]
    rng = np.random.default_rng(4)
    n = 32000
    differ = 0
    draws = 0
    for temperature, minp in CASES + [(0.0, 0.1), (1.0, 1.0)]:
        smp = abi.Sampler(n, 0xC0FFEE, temperature, minp)
        state = 0xC0FFEE
        for _ in range(25):
            lg = logits_like_a_model(rng, n, 1.5)
            tok, state = oracle.sample(lg, temperature, minp, state)
            scratch = lg.copy()  # the reference overwrites its argument (src/sampler.c:56)
            ref_tok = R.sample(C.byref(smp), scratch.ctypes.data_as(C.POINTER(C.c_float)))
            assert smp.rng_state == state
            draws += 1
            differ += ref_tok != tok
    assert differ <= draws // 100, (differ, draws)


@pytest.mark.gpu
def test_device_sampler_matches_the_oracle(hiplib):
    rng = np.random.default_rng(5)
    differ = draws = 0
    for n in (1000, 32000, 128256):
        for temperature, minp in CASES:
            state = 0xABCDEF01 + n
            for _ in range(12):
                lg = logits_like_a_model(rng, n, 1.5)
                want, nxt = oracle.sample(lg, temperature, minp, state)
                st = C.c_ulonglong(state)
                got = hiplib.calm_hip_test_sample(lg.ctypes.data_as(C.POINTER(C.c_float)), n, temperature, minp, C.byref(st))
                assert st.value == nxt
                assert 0 <= got < n
                draws += 1
                if got != want:
                    coin, _ = coin_of(state)
                    _, margin = numpy_minp(lg, temperature, minp, float(coin))
                    assert margin < 1e-5, (n, temperature, minp, got, want, margin)
                    differ += 1
                state = nxt
    assert differ <= draws // 100, (differ, draws)


@pytest.mark.gpu
def test_device_sampler_degenerate_inputs(hiplib):
    fp = C.POINTER(C.c_float)
    st = C.c_ulonglong(99)
    one_hot = np.full(5000, -50.0, dtype=np.float32)
    one_hot[4321] = 10.0
    assert hiplib.calm_hip_test_sample(one_hot.ctypes.data_as(fp), 5000, 1.0, 0.01, C.byref(st)) == 4321
    flat = np.zeros(4096, dtype=np.float32)  # everything survives; the draw is coin * n
    state = 7
    for _ in range(8):
        want, nxt = oracle.sample(flat, 1.0, 0.5, state)
        st = C.c_ulonglong(state)
        got = hiplib.calm_hip_test_sample(flat.ctypes.data_as(fp), 4096, 1.0, 0.5, C.byref(st))
        assert abs(got - want) <= 1 and st.value == nxt
        state = nxt
    short = np.array([0.1, 0.2, 0.3], dtype=np.float32)  # fewer logits than threads
    want, _ = oracle.sample(short, 1.0, 0.2, 5)
    st = C.c_ulonglong(5)
    assert hiplib.calm_hip_test_sample(short.ctypes.data_as(fp), 3, 1.0, 0.2, C.byref(st)) == want


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny_fp16", "moe_fp8"])
def test_decode_sample_hip_is_the_host_loop(hiplib, case):
    """decode_sample_hip (draws chained on the device) == forward_hip + the host sampler, token by token: same tokens,
    same generator state afterwards, same last logits"""
    from calm_amd.host import HipBackend

    model, z = load_golden(case)
    c = model.config
    n = min(20, c.seq_len - 1)
    for temperature, minp in ((0.8, 0.1), (1.0, 0.02), (0.0, 0.1)):
        b = HipBackend(model)
        try:
            state, tok, host_toks = 0x5EED, 1, []
            for pos in range(n):
                lg = b.forward(tok, pos, 0).copy()
                tok, state = oracle.sample(lg, temperature, minp, state)
                host_toks.append(tok)
            last = lg
        finally:
            b.close()
        b = HipBackend(model)
        try:
            smp = abi.Sampler(c.vocab_size, 0x5EED, temperature, minp)
            toks, lg2 = b.decode_sample(1, 0, n, smp)
            assert list(toks) == host_toks, (temperature, minp)
            assert smp.rng_state == state
            np.testing.assert_allclose(lg2, last, rtol=0, atol=1e-6 * float(np.abs(last).max()))
        finally:
            b.close()
