#!/usr/bin/env python3
"""What an fp8 (e5m2) KV cache does to logits parity, measured: the bound behind tests' FP8KV_TOL.

    python tools/fp8kv_study.py [POSITIONS] [DEEP_POSITIONS]        (GPU box; prints a table, writes gpurun_out/fp8kv/table.txt)
    python tools/fp8kv_study.py --control [LAYERS] [POSITIONS]      (no GPU: the CPU checker against itself, one weight one ulp apart)

kvbits = 8 stores every K / V element as `__nv_fp8_e5m2(float)` (src/infer.cu:473-482): a 2-bit mantissa, so neighbouring codes are
12.5-25 % apart.  Both sides (HIP backend, oracle's kvbits = 8 mode = that storage on top of src/infer.c:238-267's arithmetic) round
the SAME fp32 value except for its last bits (summation order of the k / v projections), and when those last bits straddle a rounding
boundary the two caches hold neighbouring codes for that element: a 12.5-25 % difference in one element instead of 1e-7.  This
script measures (a) how often that happens per cached element, (b) what it does to the logits, on the 2-layer attention-true model
of tests/test_long_context.py decoded from position 0 over POSITIONS positions (default 2048), every position compared, and (c)
whether it grows with depth: the full 32-layer Mistral-7B fp8 shape with kvbits = 8 over DEEP_POSITIONS positions (default 48).
The same runs with kvbits = 16 give the fp16-cache floor beside it.  Test infrastructure: the oracle is the checker.
"""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "16")

from calm_amd import calmfile as cf  # noqa: E402
from calm_amd.host import HipBackend, HostModel  # noqa: E402
from oracle import oracle  # noqa: E402


def rel_err(a, ref):
    return float(np.abs(a.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def run(model, kvbits, n_pos, seed, out):
    ref = oracle.OracleBackend(model, kvbits=kvbits)
    hip = HipBackend(model, kvbits=kvbits)
    rng = np.random.default_rng(seed)
    c = model.config
    toks = rng.integers(0, c.vocab_size, size=n_pos)
    errs = np.zeros(n_pos)
    agree = 0
    t0 = time.time()
    try:
        for pos, tok in enumerate(toks):
            lr = ref.forward(int(tok), pos, 0)
            lg = hip.forward(int(tok), pos, 0)
            errs[pos] = rel_err(lg, lr)
            agree += int(np.argmax(lg)) == int(np.argmax(lr))
        flips = total = 0
        worst_row = 0
        big = 0
        if kvbits == 8:
            for layer in range(c.n_layers):
                for which in (0, 1):
                    want = ref.kv(layer, which)[:n_pos].view(np.uint16) >> 8
                    got = hip.read_kv(layer, which)[:n_pos].view(np.uint16) >> 8
                    d = want != got
                    flips += int(d.sum())
                    total += d.size
                    worst_row = max(worst_row, int(d.sum(axis=1).max()))
                    if d.any():
                        big += int((np.abs(want[d].astype(int) - got[d].astype(int)) > 1).sum())
    finally:
        hip.close()
        ref.close()
    q = np.quantile(errs, [0.5, 0.9, 0.99, 0.999])
    line = (f"{c.n_layers:2d} layers, kvbits {kvbits:2d}, {n_pos} positions ({time.time() - t0:.0f} s): max|d|/max|logit| median {q[0]:.2e}  p90 {q[1]:.2e}  p99 {q[2]:.2e}  "
            f"p99.9 {q[3]:.2e}  max {errs.max():.2e} (at {int(errs.argmax())}); > 1e-3 at {int((errs > 1e-3).sum())} positions, > 2e-3 at {int((errs > 2e-3).sum())}; "
            f"argmax equal at {agree}/{n_pos}")
    if kvbits == 8:
        line += (f"; cached codes differing: {flips} of {total} ({flips / total:.2e} per element; most in one cached row: {worst_row} of {c.head_dim * c.n_kv_heads}); "
                 f"differences of more than one code: {big}")
    print(line, flush=True)
    out.append(line)
    # where the large ones sit: by kv length
    if kvbits == 8:
        edges = [0, 16, 64, 256, 1024, 1 << 30]
        parts = []
        for a, b in zip(edges[:-1], edges[1:]):
            e = errs[a:min(b, n_pos)]
            if e.size:
                parts.append(f"[{a},{min(b, n_pos)}): max {e.max():.2e}")
        line = "    by position: " + "  ".join(parts)
        print(line, flush=True)
        out.append(line)
    return errs


def control(layers, n_pos):
    """no GPU: the CPU checker against ITSELF with ONE norm weight of layer 0 moved by one ulp -- how far does the smallest possible
    difference between two correct implementations move the logits, with an e5m2 cache and with an fp16 one?"""
    spec = cf.SPECS["mistral-7b"]
    tensors, md = cf.synth_model_big(spec, "fp8", 1, layers)
    t2 = dict(tensors)
    w = tensors["model.layers.0.attn.norm.weight"].copy()
    w[0] = np.nextafter(w[0], np.float32(2.0))
    t2["model.layers.0.attn.norm.weight"] = w
    ma, mb = HostModel(tensors, md, context=1024), HostModel(t2, md, context=1024)
    toks = np.random.default_rng(33).integers(0, spec.vocab_size, size=n_pos)
    for kvbits in (8, 16):
        a, b = oracle.OracleBackend(ma, kvbits=kvbits), oracle.OracleBackend(mb, kvbits=kvbits)
        errs = []
        for pos, tok in enumerate(toks):
            errs.append(rel_err(b.forward(int(tok), pos, 0), a.forward(int(tok), pos, 0)))
        a.close(), b.close()
        e = np.array(errs)
        print(f"{layers:2d} layers, kvbits {kvbits:2d}: oracle vs oracle with one norm weight of layer 0 one ulp up, {n_pos} positions: "
              f"median {np.median(e):.2e}  max {e.max():.2e}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--control":
        return control(int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 24)
    n_pos = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n_deep = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    out = []
    m = cf.SPECS["mistral-7b"]
    spec = dataclasses.replace(m, name="mistral-attn", hidden_dim=4096, vocab_size=4096, n_layers=2, max_seq_len=max(n_pos, 1024))
    for seed in (31, 32):
        tensors, md = cf.synth_model_big(spec, "fp8", seed)
        model = HostModel(tensors, md, context=max(n_pos, 1024))
        for kvbits in (8, 16):
            run(model, kvbits, n_pos, seed, out)
    if n_deep > 0:
        tensors, md = cf.synth_model_big(m, "fp8", 1)
        model = HostModel(tensors, md, context=1024)
        for kvbits in (8, 16):
            run(model, kvbits, n_deep, 33, out)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "fp8kv"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fp8kv", "table.txt"), "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
