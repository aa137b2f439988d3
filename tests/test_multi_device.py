"""The layer pipeline INSIDE the library (CALM_HIP_DEVICES = P): one process, P stages, the four backend
entry points unchanged -- so the reference's own CLI drives a sharded model.  Stage s lives on device s % visible devices: with
one GPU every stage shares it (each on its own stream, the residual stream crossing by hipMemcpyPeerAsync + an event), which
exercises the whole path -- deferred uploads routed by prepare_hip, per-stage graphs, hand-offs -- and must reproduce the
single-device logits BIT FOR BIT (same kernels, same order).  The test that needs two physical devices skips itself here."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden
from calm_amd import abi
from calm_amd.host import HipBackend, load_lib
from oracle import oracle

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent(
    """
    import json, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["CALM_ROOT"])
    from calm_amd import abi
    from calm_amd.host import HipBackend, HostModel
    model = HostModel.from_file(os.environ["CALM_MODEL"])
    toks = json.loads(os.environ["CALM_TOKENS"])
    b = HipBackend(model)
    out = []
    for pos, tok in enumerate(toks):
        if pos % 5 == 3 and pos < model.config.seq_len:  # (past seq_len a step is not idempotent: it advances the sink keys)
            b.forward(tok, pos, abi.FF_UPDATE_KV_ONLY)   # a KV-only step in between (run.c's prompt steps)
            b.forward(tok, pos, 0)
        out.append(b.forward(tok, pos, 0).copy())
    np.save(os.environ["CALM_OUT"], np.stack(out))
    sys.stdout.write(json.dumps({"stages": b.stages, "devices": b.lib.calm_hip_device_count(),
                                 "stage_devices": [b.lib.calm_hip_query(b"stage_device", s) for s in range(b.stages)]}) + "\\n")
    b.close()
    """
)


def run_worker(case, stages, tmp_path):
    _, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    out = str(tmp_path / f"{case}_{stages}.npy")
    env = dict(os.environ, CALM_ROOT=ROOT, CALM_MODEL=os.path.join(GOLDEN, case + ".calm"), CALM_TOKENS=json.dumps(toks), CALM_OUT=out, CALM_HIP_DEVICES=str(stages))
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["stages"] == stages
    return np.load(out), info


@pytest.mark.parametrize("case,stages", [("tiny_fp8", 2), ("moe_fp8", 2), ("bias_tied_gf4", 2), ("sink_fp16", 2), ("dbrx_like_fp8", 2), ("tiny_fp16", 1)])
def test_stages_in_one_process_reproduce_the_single_device_logits(hiplib, tmp_path, case, stages):
    model, z = load_golden(case)
    if model.config.n_layers < stages:
        pytest.skip("fewer layers than stages")
    got, _ = run_worker(case, stages, tmp_path)
    b = HipBackend(model)
    try:
        for pos, tok in enumerate(int(t) for t in z["tokens"]):
            want = b.forward(tok, pos, 0)
            assert np.array_equal(got[pos], want), (case, stages, pos, float(np.abs(got[pos] - want).max()))
    finally:
        b.close()


KVONLY_WORKER = textwrap.dedent(
    """
    import json, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["CALM_ROOT"])
    from calm_amd import abi
    from calm_amd.host import HipBackend, HostModel
    model = HostModel.from_file(os.environ["CALM_MODEL"])
    toks = json.loads(os.environ["CALM_TOKENS"])
    out = []
    for n in json.loads(os.environ["CALM_CUTS"]):
        b = HipBackend(model, kvbits=int(os.environ["CALM_KVBITS"]))
        for pos in range(n):  # run.c's prompt loop: enqueued, not synchronised (src/run.c:208,216-218)
            b.forward(toks[pos], pos, abi.FF_UPDATE_KV_ONLY)
        out.append(b.forward(toks[n], n, 0).copy())
        b.close()
    np.save(os.environ["CALM_OUT"], np.stack(out))
    """
)


def check_kv_only_runs(case, kvbits, tmp_path):
    model, z = load_golden(case)
    if model.config.n_layers < 2:
        pytest.skip("fewer layers than stages")
    toks = [int(t) for t in z["tokens"]]
    cuts = [n for n in (3, 8, 16, 17, 20, len(toks) - 1) if n < len(toks)]
    out = str(tmp_path / f"kv_{case}.npy")
    env = dict(os.environ, CALM_ROOT=ROOT, CALM_MODEL=os.path.join(GOLDEN, case + ".calm"), CALM_TOKENS=json.dumps(toks), CALM_OUT=out, CALM_CUTS=json.dumps(cuts),
               CALM_HIP_DEVICES="2", CALM_KVBITS=str(kvbits))
    r = subprocess.run([sys.executable, "-c", KVONLY_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    for i, n in enumerate(cuts):
        if kvbits == 16:
            assert np.abs(got[i] - z["logits"][n]).max() < 1e-3 * np.abs(z["logits"][n]).max(), (case, n)
        b = HipBackend(model, kvbits=kvbits)
        try:
            for pos in range(n):
                b.forward(toks[pos], pos, abi.FF_UPDATE_KV_ONLY)
            assert np.array_equal(got[i], b.forward(toks[n], n, 0)), (case, n)
        finally:
            b.close()


@pytest.mark.parametrize("case,kvbits", [("sink_fp16", 16), ("tiny_fp8", 16), ("moe_fp8", 16), ("hd256_sink_fp8", 8), ("tiny_fp16", 8)])
def test_runs_of_kv_only_steps_on_a_sharded_model(hiplib, tmp_path, case, kvbits):
    """FF_UPDATE_KV_ONLY steps are only enqueued: nothing but the stages' own events keeps stage 0 from running ahead into stage
    1's residual stream.  Prompts of several lengths (past seq_len for the sink model) fed as runs of KV-only steps, then one
    step with logits: the golden logits of the reference, and the unsharded backend bit for bit."""
    check_kv_only_runs(case, kvbits, tmp_path)


PREFILL_WORKER = textwrap.dedent(
    """
    import json, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["CALM_ROOT"])
    from calm_amd.host import HipBackend, HostModel
    model = HostModel.from_file(os.environ["CALM_MODEL"])
    toks = json.loads(os.environ["CALM_TOKENS"])
    n = int(os.environ["CALM_NPROMPT"])
    b = HipBackend(model)
    b.prefill(toks[:n // 2], 0)              # two calls, the second at pos > 0
    b.prefill(toks[n // 2:n], n // 2)
    after = b.forward(toks[n], n, 0).copy()  # one decode step on top of the batched prompt
    b2 = HipBackend(model)
    lp = b2.prefill_logprobs(toks[:n + 1], 0)
    np.savez(os.environ["CALM_OUT"], after=after, lp=lp)
    sys.stdout.write(json.dumps({"stages": b.stages}) + "\\n")
    b.close()
    b2.close()
    """
)


def check_prompt_ingestion(case, tmp_path):
    model, z = load_golden(case)
    toks = [int(t) for t in z["tokens"]]
    n = len(toks) - 1
    if model.config.n_layers < 2:
        pytest.skip("fewer layers than stages")
    out = str(tmp_path / f"pf_{case}.npz")
    env = dict(os.environ, CALM_ROOT=ROOT, CALM_MODEL=os.path.join(GOLDEN, case + ".calm"), CALM_TOKENS=json.dumps(toks), CALM_OUT=out, CALM_NPROMPT=str(n),
               CALM_HIP_DEVICES="2")
    r = subprocess.run([sys.executable, "-c", PREFILL_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["stages"] == 2
    got = np.load(out)
    b = HipBackend(model)
    b2 = HipBackend(model)
    try:
        b.prefill(toks[: n // 2], 0)
        b.prefill(toks[n // 2 : n], n // 2)
        assert np.array_equal(got["after"], b.forward(toks[n], n, 0))
        assert np.array_equal(got["lp"], b2.prefill_logprobs(toks[: n + 1], 0))
    finally:
        b.close()
        b2.close()


@pytest.mark.parametrize("case", ["tiny_fp8", "moe_fp8", "sink_fp16", "dbrx_like_fp8"])
def test_prompt_ingestion_on_a_sharded_model(hiplib, tmp_path, case):
    """prefill_hip / prefill_logprobs_hip on CALM_HIP_DEVICES=2: a chunk runs stage after stage, the residual rows crossing like
    one token's x does; the logits of the next decode step and every scored log-probability equal the unsharded backend's bit
    for bit (sink_fp16: the prompt runs past seq_len, the tail goes through the sharded decode path inside the call)"""
    check_prompt_ingestion(case, tmp_path)


@pytest.mark.skipif(not os.path.exists(oracle.RUN_HIP), reason="oracle/_ref/run_hip (reference CLI linked to libcalm_hip.so) not built")
def test_reference_cli_drives_a_two_stage_model():
    """the unmodified run.c on CALM_HIP_DEVICES=2: it uploads tensors without knowing their layer (src/run.c:550-561), prepare_hip
    routes them; same text as on its own CPU backend"""
    env = dict(os.environ, CALM_HIP_DEVICES="2")
    env.pop("CALM_CPU", None)
    r = subprocess.run([oracle.RUN_HIP, os.path.join(GOLDEN, "tiny_fp16.calm"), "-i", "abc abc", "-t", "0", "-n", "32"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.splitlines()[1] + "\n" == open(os.path.join(GOLDEN, "cli_tiny_fp16.txt")).read()


def test_stages_on_two_physical_devices(hiplib, tmp_path):
    """needs >= 2 GPUs (self-skips on the 1-GPU box): the hand-off really crosses xGMI -- decode steps with KV-only steps in between,
    runs of KV-only steps (stage 0 must not run ahead into stage 1's input across devices either), prompt ingestion and scoring, a
    mixture-of-experts model; everything bit-equal to one device.  Peer access is enabled in both directions by init_hip."""
    if load_lib().calm_hip_device_count() < 2:
        pytest.skip("one GPU visible: the same path runs with both stages on it (tests above)")
    for case in ("tiny_fp8", "moe_fp8"):
        model, z = load_golden(case)
        got, info = run_worker(case, 2, tmp_path)
        assert info["devices"] >= 2 and info["stage_devices"][0] != info["stage_devices"][1], info
        b = HipBackend(model)
        try:
            for pos, tok in enumerate(int(t) for t in z["tokens"]):
                assert np.array_equal(got[pos], b.forward(tok, pos, 0))
        finally:
            b.close()
    check_kv_only_runs("sink_fp16", 16, tmp_path)
    check_kv_only_runs("hd256_sink_fp8", 8, tmp_path)
    check_prompt_ingestion("tiny_fp8", tmp_path)
    check_prompt_ingestion("dbrx_like_fp8", tmp_path)
