#!/usr/bin/env python3
"""How well-conditioned is a decode step of the outlier fixture (calmfile.synth_model_big(outliers=True))?  The CPU reference
(oracle/_ref = src/infer.c; our restatement if absent) against ITSELF with a few norm weights moved by one ulp -- the control that tells
a rounding-level disagreement between two correct fp32 implementations from a defect: what one ulp on the input does to the logits is
what a different summation order may do too.      python tools/outlier_conditioning.py [model] [dtype] [layers] [positions]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd import calmfile as cf
from calm_amd.host import HostModel
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
L = int(sys.argv[3]) if len(sys.argv) > 3 else cf.SPECS[name].n_layers
npos = int(sys.argv[4]) if len(sys.argv) > 4 else 4
spec = cf.SPECS[name]
for outliers in (False, True):
    tensors, md = cf.synth_model_big(spec, dtype, 5, n_layers=L, outliers=outliers)
    model = HostModel(tensors, md, context=64)
    mk = (lambda m: oracle.RefBackend(m)) if oracle.have_ref() else (lambda m: oracle.OracleBackend(m))

    def run():
        be = mk(model)
        out, tok = [], 11
        for pos in range(npos):
            lg = be.forward(tok, pos, 0).copy()
            out.append(lg)
            tok = int(np.argmax(lg))
        be.close()
        return out

    base = run()
    rows = []
    for trial in range(3):
        rng = np.random.default_rng(trial)
        saved = {}
        for l in rng.choice(L, size=min(4, L), replace=False):
            g = tensors[f"model.layers.{l}.attn.norm.weight"]
            j = int(rng.integers(0, spec.dim))
            saved[(l, j)] = g[j]
            g[j] = np.nextafter(g[j], np.float32(np.inf))
        pert = run()
        for (l, j), v in saved.items():
            tensors[f"model.layers.{l}.attn.norm.weight"][j] = v
        rows.append([float(np.abs(a - b).max() / np.abs(a).max()) for a, b in zip(base, pert)])
    print(f"{name} {dtype} L={L} outliers={outliers}: max|logit| {[round(float(np.abs(a).max()), 1) for a in base]}; "
          f"reference vs itself with 4 norm weights one ulp up, max|d|/max|logit| per position, 3 trials: " + "; ".join(" ".join(f"{e:.1e}" for e in r) for r in rows), flush=True)
