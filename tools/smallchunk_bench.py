#!/usr/bin/env python3
"""Prompt ingestion in chunks of 3 .. 16 tokens (prefill_hip called with n tokens at a time) on a layer-reduced BASELINE shape:
us per chunk and per layer with k_pf_skinny (chunks up to 8 tokens) and with the GEMM forms (knob pf_forms = 32), against n serial
decode steps.  usage: smallchunk_bench.py [model] [dtype] [layers]"""
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
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
spec = cf.SPECS[name]
model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
be = HipBackend(model, stream=cf.synth_stream_big(spec, dtype, 1, L))
rng = np.random.default_rng(0)
toks = [int(t) for t in rng.integers(0, spec.vocab_size, size=4096)]
be.prefill(toks[:64], 0)
for pos in range(16):
    be.forward(toks[pos], 64 + pos, abi.FF_UPDATE_KV_ONLY)
t0 = time.perf_counter()
for pos in range(64):
    be.forward(toks[pos], 100 + pos, abi.FF_UPDATE_KV_ONLY)
be.forward(toks[0], 164, 0)
serial = (time.perf_counter() - t0) / 65
print(f"{name} {dtype} L={L}: one serial decode step {serial*1e6:7.1f} us = {serial/L*1e6:6.2f} us per layer")
for n in (3, 4, 5, 8, 12, 16):
    row = []
    for knob in (1, 0):
        be.lib.calm_hip_configure(b"pf_forms", 0 if knob else 32)
        be.prefill(toks[:n], 200)
        reps = 40
        t0 = time.perf_counter()
        for r in range(reps):
            be.prefill(toks[r : r + n], 200 + n * (r % 8))
        dt = (time.perf_counter() - t0) / reps
        row.append(f"{'skinny' if knob else 'GEMMs '}: {dt*1e6:8.1f} us per chunk = {dt/L*1e6:7.2f} us per layer")
    be.lib.calm_hip_configure(b"pf_forms", 0)
    print(f"chunk of {n:2d} tokens: " + " | ".join(row) + f" | {n} serial steps {n*serial/L*1e6:7.2f} us per layer", flush=True)
be.close()
