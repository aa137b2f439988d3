"""Layer-pipeline sharding of the decode step across GPUs: one process per GPU, one token in flight
(BASELINE config 5).

Stage r of P owns layers [r*L/P, (r+1)*L/P) and that slice of the KV cache; the only cross-stage state is
the residual stream x (dim fp32: 16-24 KB), so there is exactly ONE point-to-point exchange per stage
boundary per token (torch.distributed send/recv: RCCL over xGMI on GPUs, gloo in the CPU tests) plus the
sampled token going back to every stage.  There is no throughput gain -- each GPU idles (P-1)/P of the
time -- it buys capacity for models beyond one GPU's 288 GB; every BASELINE model fits one MI355X, which is
why bench.py scales with replicas instead.

The compute of a stage goes through the same C ABI as everything else: forward_stage_hip on a `struct
Transformer` that describes only the stage's layers (include/calm_hip.h).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np

from .host import HostModel

STAGE_FIRST, STAGE_LAST = 1, 2


def layer_split(n_layers: int, world: int) -> List[Tuple[int, int]]:
    """contiguous, near-equal layer ranges; earlier stages take the remainder"""
    base, rem = divmod(n_layers, world)
    out, l = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((l, l + n))
        l += n
    return out


def stage_tensors(tensors: Dict[str, np.ndarray], l0: int, l1: int, first: bool, last: bool, tied: Optional[bool] = None) -> Dict[str, np.ndarray]:
    """the tensors stage [l0, l1) needs, layers renumbered from 0.  tied: the classifier shares the embedding table
    (src/run.c:112-116) -- derived from `tensors` when that is the WHOLE model, passed in when it is a slice of it"""
    out: Dict[str, np.ndarray] = {}
    if tied is None:
        tied = "model.output.weight" not in tensors
    for name, a in tensors.items():
        if name.startswith("model.layers."):
            l = int(name.split(".")[2])
            if l0 <= l < l1:
                out[name.replace(f"model.layers.{l}.", f"model.layers.{l - l0}.", 1)] = a
        elif name == "model.embed.weight":
            if first or (last and tied):
                out[name] = a
        elif name in ("model.norm.weight", "model.output.weight"):
            if last:
                out[name] = a
        elif not name.startswith("model."):
            out[name] = a
    return out


def stage_model(model: HostModel, rank: int, world: int) -> Tuple[HostModel, int]:
    """-> (HostModel of this rank's stage, stage_flags)"""
    l0, l1 = layer_split(model.config.n_layers, world)[rank]
    first, last = rank == 0, rank == world - 1
    md = dict(model.metadata)
    md["n_layers"] = str(l1 - l0)
    sm = HostModel(stage_tensors(model.tensors, l0, l1, first, last), md, context=model.config.seq_len)
    return sm, (STAGE_FIRST if first else 0) | (STAGE_LAST if last else 0)


class PipelineStage:
    """one rank of the pipeline.  `backend` offers forward_stage / export_x / import_x (calm_amd.host.HipBackend
    on a GPU; an oracle-backed stand-in in the CPU tests); `dist` is torch.distributed, already initialised."""

    def __init__(self, backend, dim: int, stage_flags: int, dist, device: str):
        import torch

        self.b = backend
        self.flags = stage_flags
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.xbuf = torch.zeros(dim, dtype=torch.float32, device=device)
        self.tok = torch.zeros(1, dtype=torch.int64, device=device)
        self._sync = torch.cuda.synchronize if device != "cpu" else (lambda: None)

    def step(self, token: int, pos: int) -> Optional[np.ndarray]:
        """one decode step of the whole pipeline seen from this rank: receive x, run the stage, send x on"""
        if not (self.flags & STAGE_FIRST):
            self.dist.recv(self.xbuf, src=self.rank - 1)
            self._sync()
            self.b.import_x(self.xbuf.data_ptr())
        logits = self.b.forward_stage(token, pos, 0, self.flags)
        if not (self.flags & STAGE_LAST):
            self.b.export_x(self.xbuf.data_ptr())
            self.dist.send(self.xbuf, dst=self.rank + 1)
        return logits

    def generate(self, first_token: int, steps: int) -> List[int]:
        """greedy decode; every rank returns the same token list (the last stage samples and broadcasts)"""
        out: List[int] = []
        token = first_token
        for pos in range(steps):
            logits = self.step(token, pos)
            if self.flags & STAGE_LAST:
                self.tok[0] = int(np.argmax(logits))
            self.dist.broadcast(self.tok, src=self.world - 1)
            self._sync()
            token = int(self.tok.item())
            out.append(token)
        return out


def main():
    """torchrun entry point: `python -m torch.distributed.run --nproc-per-node P -m calm_amd.pipeline
    --model dbrx-132b --dtype fp8 [--layers N] [--steps K]` -- a P-stage pipeline over RCCL, one rank per GPU,
    synthetic weights streamed to each stage; rank 0 prints one JSON line (tok/s of the whole pipeline)."""
    import argparse
    import json
    import os
    import time

    import torch
    import torch.distributed as dist

    from . import calmfile as cf
    from .host import HipBackend, require_torch_first

    require_torch_first("calm_amd.pipeline")  # (torch's device comes up below, before the first HipBackend)
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dbrx-132b")
    ap.add_argument("--dtype", default="fp8")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("CALM_HIP_DEVICE", str(local_rank))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank, world = dist.get_rank(), dist.get_world_size()

    spec = cf.SPECS[args.model]
    L = args.layers or spec.n_layers
    full = HostModel(cf.stub_tensors(spec, args.dtype, L), dataclasses.replace(spec, n_layers=L).metadata(args.dtype))
    sm, flags = stage_model(full, rank, world)
    l0, l1 = layer_split(L, world)[rank]

    def stream():  # this stage's share of the (deterministic) synthetic tensor stream, renumbered
        for name, a in cf.synth_stream_big(spec, args.dtype, args.seed, L):
            got = stage_tensors({name: a}, l0, l1, rank == 0, rank == world - 1, tied=spec.tied)
            for n2, a2 in got.items():
                yield n2, a2

    be = HipBackend(sm, stream=stream())
    stage = PipelineStage(be, sm.config.dim, flags, dist, "cuda")
    stage.generate(17, 8)  # warm-up: graphs captured, RCCL channels up
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks = stage.generate(17, args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "pipeline decode tok/s (batch=1, one token in flight)", "value": round(args.steps / dt, 2), "unit": "tok/s",
                          "n_gpus": world, "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "config": {"workload": f"{args.model} {args.dtype}, {L} layers over {world} stages {layer_split(L, world)}", "tokens_head": toks[:8]}}), flush=True)
    be.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
