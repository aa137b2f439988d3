#!/usr/bin/env python3
"""A/B of the chained-launch decode step (CALM_HIP_CHAIN, kernels.hip.h: ChainArgs) on the GPU box: a layer-reduced
model of a BASELINE shape, 256 greedy tokens through forward_hip + host argmax, graph replay vs eager vs chained eager.
The token streams of all three must be identical.

    python tools/chain_ab.py [model] [dtype] [layers]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel, generate, load_lib

name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
spec = cf.SPECS[name]
lib = load_lib()
model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
be = HipBackend(model, stream=cf.synth_stream_big(spec, dtype, 1, L))
ref = None
# bpc = resident 256-thread workgroups per CU the grids are sized for: a chained successor can only start early if its
# workgroups FIT beside the producer's (VGPRs: two k_ffn_down workgroups fill a CU), so 1 may beat 2 here
for label, graph, chain, bpc in (("graph replay", 1, 0, 2), ("eager", 0, 0, 2), ("eager, chained matvecs", 0, 1, 2), ("eager, chained + attention", 0, 2, 2),
                                 ("eager, chained matvecs", 0, 1, 1), ("eager, chained + attention", 0, 2, 1), ("graph replay", 1, 0, 2),
                                 ("eager, chained + attention", 0, 2, 2), ("eager, chained + attention", 0, 2, 1)):
    lib.calm_hip_configure(b"graph", graph)
    lib.calm_hip_configure(b"chain", chain)
    lib.calm_hip_configure(b"bpc", bpc)
    generate(be, model, [17], 16)
    t0 = time.perf_counter()
    toks, st = generate(be, model, [17], 256)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = toks
    print(f"{name} {dtype} L={L} {label:26s} bpc={bpc}: {256/dt:8.1f} tok/s  {dt/256*1e6:7.1f} us/token  {st['GBps']:6.0f} GB/s  same tokens: {toks == ref}", flush=True)
lib.calm_hip_configure(b"graph", 1)
lib.calm_hip_configure(b"chain", 0)
lib.calm_hip_configure(b"bpc", 2)
be.close()
