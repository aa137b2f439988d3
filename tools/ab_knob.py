#!/usr/bin/env python3
"""A/B of a launch-shaping knob of the library (calm_hip_configure(key, v); KNOB_VALUES=0,1 by default) on a layer-reduced BASELINE
shape: per-stage timings, tok/s over 256 greedy tokens, identical tokens.
    python tools/ab_knob.py <knob> [model] [dtype] [layers]      e.g.  qkv_attn mistral-7b fp8 8"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd import calmfile as cf
from calm_amd.host import STAGES, HipBackend, HostModel, generate, load_lib

knob = sys.argv[1].encode()
name = sys.argv[2] if len(sys.argv) > 2 else "mistral-7b"
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp8"
L = int(sys.argv[4]) if len(sys.argv) > 4 else 8
spec = cf.SPECS.get(name) or cf.ARCH_SPECS[name]
lib = load_lib()
model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
be = HipBackend(model, device_synth=(spec, dtype, 1, L))
ref = None
for rep in range(2):
    for v in [int(x) for x in os.environ.get("KNOB_VALUES", "0,1").split(",")]:
        lib.calm_hip_configure(knob, v)
        generate(be, model, [17], 32)
        row = " | ".join(f"{st} {be.stage_us(i, 8 if i != 5 else 2)[0]:6.2f}" for i, st in enumerate(STAGES))
        t0 = time.perf_counter()
        toks, st = generate(be, model, [17], 256)
        dt = time.perf_counter() - t0
        ref = ref or toks
        print(f"{knob.decode()}={v}: {row} | {256 / dt:8.1f} tok/s {dt / 256 * 1e6:7.1f} us/token  same tokens: {toks == ref}", flush=True)
be.close()
