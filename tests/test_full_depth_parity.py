"""Full-DEPTH parity of the BASELINE configurations (north_star's own clause): the whole synthetic model -- every layer, full
width, full vocabulary -- on the HIP backend against the reference CPU path (oracle/_ref = the untouched src/infer.c:311-472,
driven like src/run.c:167-256 drives it; our C restatement if that binary is absent), same seeded weights on both sides:

  (i)  logits of every decoded position, teacher-forced along the reference's own greedy stream:
       max|delta| / max|logit| <= 1e-3 per token (fp16 AND fp8 / gf4: weights decode exactly, activations stay fp32);
  (ii) the greedy stream itself: the HIP argmax equals the reference's next token at every position -- which makes the
       free-running 256-token streams identical, both sides being deterministic -- or, where it does not, the reference's own
       top-2 margin at that position is below 4 x tol x max|logit| (random-weight logits are near-tied now and then:
       SURVEY.md section 7.3 #2) and the test says where.

Error grows ~sqrt(L), so the layer-reduced tests elsewhere do not cover this.
Needs the host copy of the model for the CPU side (7.2 GB for Mistral-7B fp8) and ~40 ms of CPU per reference token.
"""
import numpy as np
import pytest

from conftest import rel_err
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel
from oracle import oracle

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(autouse=True)
def more_cpu_threads():
    """the CPU reference walks 2-8 GB of weights per token here: give it half the cores (its own default, src/infer.c:171-176;
    conftest pins 8 for the tiny models) -- through the OpenMP runtime, the environment having been read already"""
    import ctypes
    import os

    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(max(8, min(32, (os.cpu_count() or 16) // 2)))
        yield
        gomp.omp_set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
    except OSError:
        yield


# the outlier fixture (calmfile.synth_model_big(outliers=True)): what two correct fp32 evaluations of a step may differ by.  Measured against
# a float64 evaluation of position 0 (oracle/f64_step.py; profiles/r06_outliers.txt): the CPU reference itself sits 2.2e-4 ... 9.4e-4 from it,
# our restatement 1.2e-4 ... 1.1e-3, the HIP backend 0.9e-4 ... 8.7e-4 -- so a pair of them is up to 1.4e-3 apart (benign fixture: all three
# 2.3e-4 ... 4.1e-4, pairs <= 3.6e-4).  The bound below is that, with headroom; the float64 anchor is asserted separately.
OUTLIER_TOL = 2.5e-3


def full_depth_check(name, dtype, n_tokens, seed=3, first_token=11, outliers=False, tol=TOL):
    """-> dict(worst relative error, positions where argmax differed, reference tokens, hip tokens)"""
    spec = cf.SPECS[name]
    tensors, md = cf.synth_model_big(spec, dtype, seed, outliers=outliers)  # the real depth: n_layers is not overridden
    model = HostModel(tensors, md)
    assert model.config.n_layers == spec.n_layers
    ref = oracle.RefBackend(model) if oracle.have_ref() else oracle.OracleBackend(model)
    hip = HipBackend(model)
    try:
        # the reference's free-running greedy decode (run.c's loop), keeping every position's logits
        ref_tokens, ref_logits, tok = [], [], first_token
        for pos in range(n_tokens):
            lg = ref.forward(tok, pos, 0)
            ref_logits.append(lg.copy())
            tok = oracle.argmax(lg)
            ref_tokens.append(tok)
        # HIP along the same stream
        worst, diverged, hip_tokens, tok = 0.0, [], [], first_token
        for pos in range(n_tokens):
            lg = hip.forward(tok, pos, 0)
            lr = ref_logits[pos]
            assert np.isfinite(lg).all(), pos
            e = rel_err(lg, lr)
            worst = max(worst, e)
            assert e <= tol, f"{name} {dtype}: position {pos}: max|delta|/max|logit| = {e:.3e} > {tol}"
            h = int(lg.argmax())
            hip_tokens.append(h)
            if h != ref_tokens[pos]:
                top2 = np.partition(lr, -2)[-2:]
                margin = float(top2[1] - top2[0])
                diverged.append((pos, h, ref_tokens[pos], margin, float(np.abs(lr).max())))
            tok = ref_tokens[pos]
        # the same stream as a PROMPT: prefill_hip over positions 0 .. m - 1 (f16 matrix cores, hi + lo activations, every layer),
        # then the decode step at position m against the reference's logits there; and the scored log-probabilities of the stream
        m = n_tokens - 1
        prompt = [first_token] + ref_tokens[: m - 1]
        redone0 = hip.lib.calm_hip_query(b"pf_redone", 0)
        hip.prefill(prompt, 0)
        pf_err = rel_err(hip.forward(ref_tokens[m - 1], m, 0), ref_logits[m])
        lp = hip.prefill_logprobs(prompt + [ref_tokens[m - 1]], 0)
        want = []
        for pos in range(m):
            lr = ref_logits[pos].astype(np.float64)
            want.append((lr[ref_tokens[pos]] - lr.max()) - np.log(np.exp(lr - lr.max()).sum()))
        lp_err = float(np.abs(lp[:m] - np.array(want)).max())
        out = {"worst": worst, "diverged": diverged, "ref_tokens": ref_tokens, "hip_tokens": hip_tokens, "prefill": pf_err, "logprob": lp_err,
               "pf_redone": hip.lib.calm_hip_query(b"pf_redone", 0) - redone0, "logit_max": float(max(np.abs(l).max() for l in ref_logits))}
        if outliers:  # position 0 against exact arithmetic: who is further from it, the reference or the HIP backend?
            from oracle.f64_step import position0_logits_f64

            l64 = position0_logits_f64(tensors, md, spec, dtype, [first_token])[0]
            f = lambda a: float(np.abs(a.astype(np.float64) - l64).max() / np.abs(l64).max())
            out["f64"] = (f(hip.forward(first_token, 0, 0)), f(ref_logits[0]))
        return out
    finally:
        hip.close()
        ref.close()


@pytest.mark.parametrize("name,dtype,n_tokens", [("mistral-7b", "fp8", 256), ("llama-3-8b", "gf4", 64), ("tinyllama-1.1b", "fp16", 64)])
def test_full_depth_logits_and_greedy_stream_match_the_reference(hiplib, name, dtype, n_tokens):
    r = full_depth_check(name, dtype, n_tokens)
    print(f"{name} {dtype} full depth, {n_tokens} positions: worst max|d|/max|logit| = {r['worst']:.3e}; argmax differences: {r['diverged']}; "
          f"after prefill_hip of the stream {r['prefill']:.3e}; scored log-probabilities off by at most {r['logprob']:.3e}")
    assert r["prefill"] <= TOL, r["prefill"]
    assert r["logprob"] <= 5e-3, r["logprob"]  # log-probabilities of greedy picks: |d logit| <= 1e-3 max|logit| ~ a few 1e-3
    for pos, h, t, margin, lmax in r["diverged"]:
        # a different pick is only acceptable at a near-tie of the REFERENCE's own logits
        assert margin < 4 * TOL * lmax, f"{name} {dtype}: greedy streams part at position {pos} (hip {h}, reference {t}) with a reference top-2 margin of {margin:.3e}"
    if not r["diverged"]:
        assert r["hip_tokens"] == r["ref_tokens"]


@pytest.mark.parametrize("name,dtype,n_tokens,outliers", [("mistral-7b", "fp8", 48, 1), ("llama-3-8b", "gf4", 48, 1), ("mistral-7b", "fp8", 24, 2)])
def test_full_depth_parity_under_outlier_statistics(hiplib, name, dtype, n_tokens, outliers):
    """The same comparison on the fixture with a trained model's statistics (calmfile.synth_model_big(outliers=True), round 6): residual
    channels of 10^2 - 10^3, Student-t weights, log-normal norm weights -- what the reference's own quality gate (perplexity on real
    checkpoints, src/run.c:258-316) meets and N(0, 1 / fan_in) weights never produce.  It exercises the 1e-3 bound where a few channels
    carry the norm, gf4's 2^-a image scaling on large activations, and the prompt path's hi + lo binary16 split.  outliers = 2 pushes
    gated hidden activations past 65504: a chunk whose activations leave the binary16 range must be sent back through the serial path
    (pf_redone counts its tokens) -- at full size, with no loss of parity."""
    r = full_depth_check(name, dtype, n_tokens, seed=5, outliers=outliers, tol=OUTLIER_TOL)
    assert (r["pf_redone"] > 0) == (outliers == 2), r["pf_redone"]
    print(f"{name} {dtype} full depth, outlier fixture, {n_tokens} positions: worst max|d|/max|logit| = {r['worst']:.3e} (max |logit| {r['logit_max']:.1f}); argmax differences: "
          f"{r['diverged']}; after prefill_hip of the stream {r['prefill']:.3e} ({r['pf_redone']} prompt tokens redone serially); scored log-probabilities off by at most {r['logprob']:.3e}; "
          f"position 0 against float64: HIP {r['f64'][0]:.2e}, CPU reference {r['f64'][1]:.2e}")
    # the anchor: the HIP backend is about as close to exact arithmetic as the reference is
    assert r["f64"][0] <= max(2 * r["f64"][1], 1e-3), r["f64"]
    assert r["prefill"] <= OUTLIER_TOL, r["prefill"]
    assert r["logprob"] <= 5e-3 * max(1.0, r["logit_max"] / 5), r["logprob"]
    for pos, h, t, margin, lmax in r["diverged"]:
        assert margin < 4 * OUTLIER_TOL * lmax, f"{name} {dtype}: greedy streams part at position {pos} (hip {h}, reference {t}) with a reference top-2 margin of {margin:.3e}"


def test_full_depth_free_running_stream_mistral7b(hiplib):
    """the headline configuration decoded FREE-RUNNING on both sides (each feeds on its own picks), 64 tokens: identical, or the
    first difference sits on a reference near-tie (then the test reports it and stops comparing there)"""
    spec = cf.SPECS["mistral-7b"]
    tensors, md = cf.synth_model_big(spec, "fp8", 7)
    model = HostModel(tensors, md)
    ref = oracle.RefBackend(model) if oracle.have_ref() else oracle.OracleBackend(model)
    hip = HipBackend(model)
    try:
        n = 64
        dev, _ = hip.decode_greedy(23, 0, n)  # device-side loop: forward + on-device argmax, n tokens, no host in between
        tok = 23
        for pos in range(n):
            lr = ref.forward(tok, pos, 0)
            tok = oracle.argmax(lr)
            if int(dev[pos]) != tok:
                top2 = np.partition(lr, -2)[-2:]
                assert top2[1] - top2[0] < 4 * TOL * np.abs(lr).max(), (pos, int(dev[pos]), tok)
                print(f"free-running streams part at position {pos} on a reference near-tie (margin {top2[1] - top2[0]:.2e})")
                break
    finally:
        hip.close()
        ref.close()
