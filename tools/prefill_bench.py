#!/usr/bin/env python3
"""Prompt-ingestion rate on the GPU box: prefill_hip (1024-token chunks on the f16 matrix cores, hi + lo activations) against the
serial FF_UPDATE_KV_ONLY loop the reference runs (src/run.c:208,216-218), on a layer-reduced model of a
BASELINE shape; rates are per layer-reduced model and scaled to the full depth by layer count.

usage: prefill_bench.py [model] [dtype] [layers] [prompt tokens]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from calm_amd import abi
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel

name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = int(sys.argv[4]) if len(sys.argv) > 4 else 512
spec = cf.SPECS[name]
model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
be = HipBackend(model, stream=cf.synth_stream_big(spec, dtype, 1, L))
for kv_ in os.environ.get("KNOBS", "").split():  # e.g. KNOBS="attn_vt=0"
    key, val = kv_.split("=")
    assert be.lib.calm_hip_configure(key.encode(), int(val)) >= 0, key
rng = np.random.default_rng(0)
toks = [int(t) for t in rng.integers(0, spec.vocab_size, size=N)]

PROFILE = bool(os.environ.get("PF_PROFILE"))  # under rocprofv3: launches of ONE chunk size only, no serial comparison
be.prefill(toks[: N if PROFILE else 64], 0)  # warm-up (allocations, code load)
for n in ((N,) if PROFILE else (64, N)):
    t0 = time.perf_counter()
    be.prefill(toks[:n], 0)
    dt = time.perf_counter() - t0
    act = spec.n_experts_active if spec.n_experts else 1
    flop = 2.0 * n * L * (spec.dim * (spec.n_heads * spec.head_dim * 2 + 2 * spec.n_kv_heads * spec.head_dim) + 3 * act * spec.dim * spec.hidden_dim)
    print(f"prefill {n:5d} tokens, L={L}: {dt*1e3:8.2f} ms = {n/dt:9.0f} tok/s ({dt/n/L*1e6:7.2f} us/token/layer, {flop/dt/1e12:6.1f} TFLOP/s algorithmic); "
          f"full depth ({spec.n_layers} layers): {n/dt*L/spec.n_layers:8.0f} tok/s")
if PROFILE:
    be.close()
    sys.exit(0)
lg_b = be.forward(toks[N - 1], N - 1, 0).copy()

be.forward(toks[0], 0, abi.FF_UPDATE_KV_ONLY)
n = min(N, 128)
t0 = time.perf_counter()
for pos in range(n):
    be.forward(toks[pos], pos, abi.FF_UPDATE_KV_ONLY)
be.forward(toks[n - 1], n - 1, 0)  # synchronises
dt = time.perf_counter() - t0
print(f"serial  {n:5d} tokens, L={L}: {dt*1e3:8.2f} ms = {n/dt:9.0f} tok/s; full depth: {n/dt*L/spec.n_layers:8.0f} tok/s")
be.close()
