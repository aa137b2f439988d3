"""Layer-pipeline sharding (BASELINE config 5 / SURVEY 8e): the stage protocol of calm_amd/pipeline.py over
world_size 2 and 3 with gloo on CPU.  Each rank's compute is the oracle restricted to its stage (the same
forward_stage / export_x / import_x surface HipBackend offers on a GPU); the greedy token stream must equal
the unsharded oracle's, which tests/test_oracle.py pins to the reference."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from calm_amd import calmfile as cf
from calm_amd.host import HostModel
from calm_amd.pipeline import layer_split, stage_model, stage_tensors
from conftest import GOLDEN, ROOT, load_golden, run_torchrun

WORKER = textwrap.dedent(
    """
    import os, sys, json
    import torch.distributed as dist
    sys.path.insert(0, os.environ["CALM_ROOT"])
    from calm_amd.host import HostModel
    from calm_amd.pipeline import PipelineStage, stage_model
    from oracle import oracle
    dist.init_process_group("gloo")
    model = HostModel.from_file(os.environ["CALM_MODEL"])
    sm, flags = stage_model(model, dist.get_rank(), dist.get_world_size())
    stage = PipelineStage(oracle.OracleBackend(sm), sm.config.dim, flags, dist, "cpu")
    toks = stage.generate(int(os.environ["CALM_FIRST"]), int(os.environ["CALM_STEPS"]))
    # one write() per rank, newline included: the ranks share the launcher's pipe and print()'s separate newline interleaves
    sys.stdout.write(json.dumps({"rank": dist.get_rank(), "layers": sm.config.n_layers, "tokens": toks}) + "\\n")
    sys.stdout.flush()
    dist.destroy_process_group()
    """
)


def test_layer_split():
    assert layer_split(40, 4) == [(0, 10), (10, 20), (20, 30), (30, 40)]
    assert layer_split(5, 3) == [(0, 2), (2, 4), (4, 5)]
    assert layer_split(2, 2) == [(0, 1), (1, 2)]


def test_stage_tensors_renumber_and_boundaries():
    t, md = cf.synth_model(cf.tiny_spec(n_layers=3, tied=True), "fp8", seed=2)
    s0 = stage_tensors(t, 0, 2, True, False)
    s1 = stage_tensors(t, 2, 3, False, True)
    assert "model.embed.weight" in s0 and "model.norm.weight" not in s0
    assert "model.layers.1.mlp.w2.weight" in s0 and "model.layers.2.mlp.w2.weight" not in s0
    assert "model.layers.0.attn.wq.weight" in s1 and s1["model.layers.0.attn.wq.weight"] is t["model.layers.2.attn.wq.weight"]
    assert "model.norm.weight" in s1 and "model.embed.weight" in s1  # tied classifier lives in the embedding
    m = HostModel(t, md)
    sm, flags = stage_model(m, 1, 2)
    assert flags == 2 and sm.config.n_layers in (1, 2)


@pytest.mark.parametrize("case,world", [("tiny_fp8", 2), ("moe_fp8", 2), ("bias_tied_gf4", 2)])
def test_pipeline_matches_unsharded_greedy_stream(tmp_path, case, world):
    model, z = load_golden(case)
    want = [int(t) for t in z["tokens"][1:13]]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, CALM_ROOT=ROOT, CALM_MODEL=os.path.join(GOLDEN, case + ".calm"), CALM_FIRST=str(int(z["tokens"][0])), CALM_STEPS="12", OMP_NUM_THREADS="1")
    r = run_torchrun(script, world, env)
    assert r.returncode == 0, r.stderr[-3000:]
    outs = [json.loads(l) for l in r.stdout.replace("}{", "}\n{").splitlines() if l.startswith("{")]
    assert len(outs) == world and sum(o["layers"] for o in outs) == model.config.n_layers
    for o in outs:
        assert o["tokens"] == want, o
