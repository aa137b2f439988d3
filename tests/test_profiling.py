"""The profiling surface: perf_hip's per-stage breakdown -- the table the reference prints from
perf_cuda (src/infer.cu:761-801), reached through the reference's own CLI -- and the per-kernel algorithmic-byte account
(CALM_HIP_PROF_JSON) that tools/hipprof.sh joins with the rocprofv3 kernel trace, the way the reference's PROF_TOKEN feeds
tools/cudaprof.cu:85-100."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from calm_amd import calmfile as cf
from calm_amd.host import STAGES, HipBackend, HostModel
from oracle import oracle

pytestmark = pytest.mark.gpu

ROW = re.compile(r"\[(\d+)\]\s+(\w+):\s+([\d.]+)%;\s+([\d.]+) usec/run,\s+([\d.]+) GB/s")


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.skipif(not os.path.exists(oracle.RUN_HIP), reason="oracle/_ref/run_hip (reference CLI linked to libcalm_hip.so) not built")
def test_perf_hip_breakdown_through_the_reference_cli(hiplib, tmp_path, fused):
    """fused = 1: the step's short-context attention rides in k_qkv's launch (k_qkv_attn, round 6) -- ONE launch, so the table has no
    attention row and the QKV row carries both stages' bytes; fused = 0 (CALM_HIP_QKV_ATTN=0): the reference's six rows."""
    path = str(tmp_path / "mistral_l4_fp8.calm")
    spec, L, n = cf.SPECS["mistral-7b"], 4, 48
    cf.write_synth_big(path, spec, "fp8", seed=1, n_layers=L)
    acct = str(tmp_path / "kernel_bytes.json")
    env = dict(os.environ, CALM_HIP_PROF="1", CUDA_INJECTION64_PATH="1", CALM_HIP_PROF_JSON=acct, CALM_HIP_QKV_ATTN=str(fused))  # run.c:630 gates perf_*() on the injection variable
    env.pop("CALM_CPU", None)
    r = subprocess.run([oracle.RUN_HIP, path, "-i", "abc", "-t", "0", "-n", str(n)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "forward_hip breakdown" in r.stdout
    rows = {m.group(2): (float(m.group(3)), float(m.group(4)), float(m.group(5))) for m in ROW.finditer(r.stdout)}
    assert set(rows) == {"matmul_qkv", "matmul_attn", "matmul_ffn_up", "matmul_ffn_down", "output"} | (set() if fused else {"attn"}), r.stdout[-1500:]
    assert abs(sum(v[0] for v in rows.values()) - 100.0) < 0.5
    # the same stages timed back to back by perf_stage_hip on the same file
    model = HostModel.from_file(path)
    old = hiplib.calm_hip_configure(b"qkv_attn", fused)
    b = HipBackend(model)
    try:
        for pos in range(8):
            b.forward(3 + pos, pos, 0)
        names = {"qkv": "matmul_qkv", "attn_out": "matmul_attn", "ffn_up": "matmul_ffn_up", "ffn_down": "matmul_ffn_down", "output": "output"}
        seen = {}
        for i, st in enumerate(STAGES):
            if st not in names:
                continue
            us, nbytes = b.stage_us(i, 8 if st != "output" else 2)
            seen[st] = (rows[names[st]][2], round(nbytes / us / 1e3, 1), rows[names[st]][1], round(us, 2))  # GB/s table, GB/s here, us table, us here
            # the table's own arithmetic, which does not depend on how fast this box ran either measurement: GB/s x usec per run must
            # be the bytes perf_stage_hip files under this stage, times its launches per run (L layers; the classifier once) -- the byte
            # tagging is what the row is FOR (src/infer.cu:692-699: bw = bytes / time); printed with one decimal each
            # (usec/run divides by ALL forward calls of the run, and the prompt's KV-only steps launch no classifier: its row may sit
            # below by their share, three or four steps of 49)
            per_run = nbytes * (1 if st == "output" else L)
            got = rows[names[st]][2] * rows[names[st]][1] * 1e3
            if fused and st == "qkv":
                # (the fused launch's bytes include the cached rows it read, which grow with the position: perf_stage_hip quotes them at
                # the 8th position, the table averages over the run's 48)
                kv_row = 2 * spec.kv_dim * 2
                assert 0.98 * per_run <= got <= 1.02 * (per_run + L * 48 * kv_row), (st, rows[names[st]], per_run)
                continue
            assert (0.88 if st == "output" else 0.98) * per_run <= got <= 1.02 * per_run, (st, rows[names[st]], per_run)
        diag = os.environ.get("CALM_TEST_DIAG")  # the GPU sessions collect these across whole-suite runs (profiles/r05_gpu_tests.txt)
        if diag:
            with open(diag, "a") as f:
                f.write("test_profiling table-vs-stage " + json.dumps({k: [v[0], v[1], round(v[0] / v[1], 3)] for k, v in seen.items()}) + "\n")
        # ... and the two clocks agree grossly.  Round 5: whole-suite runs on fresh boxes went red in THIS test twice in four (and never
        # in four runs of this file alone) while the band below was 20 %: the CLI child decodes 15 ms of GPU work after seconds of
        # loading, perf_stage_hip times a few launches in a process that has been busy for minutes, and the clocks the two bursts meet are
        # not the same.  The band now separates a wrong unit or byte count from that noise; the ratios of every run are on record.
        for st, (table, gbps, _, _) in seen.items():
            assert 0.6 * gbps <= table <= 1.5 * gbps, (st, seen, r.stdout[-1200:])
    finally:
        b.close()
        hiplib.calm_hip_configure(b"qkv_attn", old)
    # the byte account: launches and bytes of every decode kernel of the run (warm-up step + n - 1 decode steps + prompt steps)
    a = json.load(open(acct))
    qkv = "k_qkv_attn" if fused else "k_qkv"
    assert set(a) >= {qkv, "k_attn_out", "k_ffn_up", "k_ffn_down", "k_output"} | (set() if fused else {"k_attn"})
    per = lambda k: a[k]["algorithmic_bytes"] / a[k]["launches"]
    assert per("k_ffn_up") == 2 * spec.hidden_dim * spec.dim and per("k_ffn_down") == spec.hidden_dim * spec.dim
    assert per("k_attn_out") == spec.q_dim * spec.dim
    assert per("k_output") == spec.vocab_size * spec.dim
    assert a[qkv]["launches"] == a["k_ffn_up"]["launches"] and a[qkv]["launches"] % L == 0
    steps = a[qkv]["launches"] // L
    qkv_w = (spec.q_dim + 2 * spec.kv_dim) * spec.dim
    if fused:  # the fused launch's account = the weights + the cached rows every step read (positions 0 .. steps - 1, binary16 K and V)
        assert qkv_w < per(qkv) <= qkv_w + steps * 2 * spec.kv_dim * 2
    else:
        assert per(qkv) == qkv_w
    # n_bandwidth of the reference's accounting (src/run.c:523-532) is what one step's weight kernels add up to
    weights = (qkv_w + sum(per(k) for k in ("k_attn_out", "k_ffn_up", "k_ffn_down"))) * L + per("k_output")
    assert weights == model.accounting()[2] - sum(model.tensors[nm].nbytes for nm in model.tensors if nm.endswith("norm.weight"))
    assert steps >= n - 1
