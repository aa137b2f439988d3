#!/usr/bin/env python3
"""What the short-context attention launch costs as a function of the cache length (round 5, verdict item 7): per-stage event timings
(perf_stage_hip) of k_attn on the Mistral-7B attention geometry after decoding kv_len positions, kv_len = 1 ... 384.  The intercept is
what ANY one-launch attention stage costs here before it has read a byte of the cache; the slope is what a restructured position loop
(more CUs per kv head, rows read once for the 4 query heads) could at most remove.      python tools/attn_floor.py [layers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd import calmfile as cf
from calm_amd.host import STAGES, HipBackend, HostModel, generate

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spec = cf.SPECS["mistral-7b"]
model = HostModel(cf.stub_tensors(spec, "fp8", L), cf.dataclasses.replace(spec, n_layers=L).metadata("fp8"))
be = HipBackend(model, stream=cf.synth_stream_big(spec, "fp8", 1, L))
i_attn, i_begin = STAGES.index("attn"), None
print("kv_len   k_attn us (3 runs)      neighbours at this length: qkv, attn_out us")
for kv in (1, 2, 8, 32, 64, 128, 192, 256, 320, 384):
    generate(be, model, [17], kv)
    runs = [be.stage_us(i_attn, 8)[0] for _ in range(3)]
    q = be.stage_us(STAGES.index("qkv"), 8)[0]
    o = be.stage_us(STAGES.index("attn_out"), 8)[0]
    print(f"{kv:5d}    " + "  ".join(f"{u:5.2f}" for u in runs) + f"        {q:5.2f}  {o:5.2f}", flush=True)
be.close()
