"""Multi-GPU = independent replicas of the decode path (one token in flight per sequence: the path
does not shard for models that fit one GPU's 288 GB).  The only cross-rank traffic is the timing
protocol of bench.py, factored out here so it can be tested over gloo on CPU."""
from __future__ import annotations


def aggregate_throughput(dist, steps: int, elapsed: float, device: str = "cuda"):
    """every rank ran `steps` decode steps in `elapsed` seconds (measured between barriers);
    -> whole-job rate = world * steps / max-over-ranks(elapsed)"""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    if dist is not None and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return {"world": world, "elapsed": elapsed, "value": world * steps / elapsed}
