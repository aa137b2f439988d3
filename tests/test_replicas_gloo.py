"""N > 1 in bench.py = independent replicas: no data-path collective, only the
timing protocol -- barrier, max-over-ranks of the elapsed time, whole-job tokens / that time.
Exercised here with world_size 2 over gloo on CPU (the same code path bench.py runs over RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT, run_torchrun

WORKER = textwrap.dedent(
    """
    import os, sys, time, json
    import torch, torch.distributed as dist
    sys.path.insert(0, os.environ["CALM_ROOT"])
    from calm_amd.replicas import aggregate_throughput
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    dist.barrier()
    elapsed = 0.10 if rank == 0 else 0.25   # rank 1 is the slow replica
    out = aggregate_throughput(dist, steps=50, elapsed=elapsed, device="cpu")
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()
    """
)


def test_two_rank_replica_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, CALM_ROOT=ROOT)
    r = run_torchrun(script, 2, env)
    assert r.returncode == 0, r.stderr[-3000:]
    import json

    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["world"] == 2
    assert abs(out["elapsed"] - 0.25) < 1e-9          # max over ranks
    assert abs(out["value"] - 2 * 50 / 0.25) < 1e-6   # whole-job steps / slowest rank's time
