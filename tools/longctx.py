#!/usr/bin/env python3
"""Decode rate deep into a long context (the reference README's "last 32 tokens" columns): Mistral-7B fp8 shape cut
to L layers, context 32768 with the fp8 KV cache the reference switches to beyond 4096 (src/run.c:536-540).
The cache is not filled by a real prompt -- attention cost does not depend on the values -- only its length matters."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd import calmfile as cf
from calm_amd.host import STAGES, HipBackend, HostModel, generate, load_lib

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spec = cf.SPECS[os.environ.get("MODEL", "mistral-7b")]  # MODEL=dbrx-132b: kv_mul 6 (all six query heads of a kv head share a workgroup)
lib = load_lib()
for kv_ in os.environ.get("KNOBS", "").split():  # e.g. KNOBS="split_min=100000"
    key, val = kv_.split("=")
    assert lib.calm_hip_configure(key.encode(), int(val)) >= 0, key
POS = [int(x) for x in os.environ["POS"].split(",")] if os.environ.get("POS") else None
for kvbits, ctx in ((16, 4096), (8, 32768)) if not os.environ.get("KV16_32K") else ((16, 32768),):
    md = cf.dataclasses.replace(spec, n_layers=L, max_seq_len=ctx).metadata("fp8")
    model = HostModel(cf.stub_tensors(spec, "fp8", L), md, context=ctx)
    be = HipBackend(model, kvbits=kvbits, stream=cf.synth_stream_big(spec, "fp8", 1, L))
    generate(be, model, [17], 16, kvbits=kvbits)
    for pos in (POS if POS else ([256, 4000] if ctx == 4096 else [256, 4000, 8000, 16000, 32000])):
        for split_t in [int(x) for x in os.environ.get("SPLITS", "0").split(",")]:  # 0 = the backend's default
            old = lib.calm_hip_configure(b"split_t", split_t) if split_t else None
            generate(be, model, [17], 8, pos_offset=pos, kvbits=kvbits)
            t0 = time.perf_counter()
            _, st = generate(be, model, [17], 32, pos_offset=pos + 8, kvbits=kvbits)
            dt = time.perf_counter() - t0
            us, b = be.stage_us(1, 6)
            kv_mb = 2 * (kvbits // 8) * spec.n_kv_heads * spec.head_dim * (pos + 40) / 1e6
            tag = f" split_t {split_t:3d}" if split_t else ""
            print(f"kv{kvbits:2d} ctx {ctx:5d} pos ~{pos:5d}{tag}: {dt/32*1e6:8.1f} us/token (L={L}); attention stage {us:7.2f} us for {kv_mb:6.1f} MB/layer = {kv_mb/us*1e3:6.0f} GB/s; "
                  f"full depth ~{32/dt*L/32:7.1f} tok/s", flush=True)
            if split_t:
                lib.calm_hip_configure(b"split_t", old)
    be.close()
