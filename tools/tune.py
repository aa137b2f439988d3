#!/usr/bin/env python3
"""Stage-time sweep on the GPU box: a Mistral-7B-shaped model cut to 8 layers (1.7 GB of layer weights,
far beyond the 256 MiB Infinity Cache), per-stage event timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from calm_amd import calmfile as cf
from calm_amd.host import STAGES, HipBackend, HostModel, generate, load_lib

name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
brief = len(sys.argv) > 4
spec = cf.SPECS.get(name) or cf.ARCH_SPECS[name]
lib = load_lib()
for kv_ in os.environ.get("KNOBS", "").split():  # e.g. KNOBS="forms=1 qkv_attn=0"
    key, val = kv_.split("=")
    assert lib.calm_hip_configure(key.encode(), int(val)) >= 0, key
model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
be = HipBackend(model, stream=cf.synth_stream_big(spec, dtype, 1, L))
for kv in ((256,) if brief else (128, 256)):
    generate(be, model, [17], kv)
    print(f"== {name} {dtype} L={L} kv_len={kv}")
    row = []
    for i, st in enumerate(STAGES):
        us, b = be.stage_us(i, 6 if i != 5 else 2)
        row.append(f"{st} {us:6.2f}us {b/us/1e3:6.0f}GB/s")
    print(" | ".join(row), flush=True)
if not brief or os.environ.get("LONGCTX"):
    # long context: the last 32 positions of a 4096 window (the reference README's "last 32" column)
    generate(be, model, [17], 8, pos_offset=4000)
    t0 = time.perf_counter()
    _, st = generate(be, model, [17], 32, pos_offset=4008)
    dt = time.perf_counter() - t0
    us, b = be.stage_us(1, 6)
    print(f"last-32 @pos~4040: {32/dt:8.1f} tok/s ({dt/32*1e6:7.1f} us/token, L={L}); attn stage {us:6.2f}us {b/us/1e3:6.0f}GB/s")
    for st_ in (512, 256, 64):
        old = lib.calm_hip_configure(b"split_t", st_)
        us, b = be.stage_us(1, 6)
        t0 = time.perf_counter()
        generate(be, model, [17], 32, pos_offset=4008)
        dt = time.perf_counter() - t0
        print(f"   split_t={st_:4d}: attn stage {us:6.2f}us {b/us/1e3:6.0f}GB/s; last-32 {32/dt:8.1f} tok/s")
        lib.calm_hip_configure(b"split_t", old)
for graph in ((1,) if brief else (1, 0)):
    lib.calm_hip_configure(b"graph", graph)
    generate(be, model, [17], 16)
    t0 = time.perf_counter()
    toks, st = generate(be, model, [17], 256)
    dt = time.perf_counter() - t0
    print(f"graph={graph}: {256/dt:8.1f} tok/s, {dt/256*1e6:7.1f} us/token, {st['GBps']:.0f} GB/s (L={L})")
lib.calm_hip_configure(b"graph", 1)
be.close()
