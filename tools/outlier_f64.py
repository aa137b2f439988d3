#!/usr/bin/env python3
"""Who is off at position 0 of the outlier fixture?  A float64 restatement of the first decode step (one cached position: the attention
output IS the value row; src/infer.c:311-472 otherwise) against (a) the CPU reference and (b) the HIP backend's logits saved by
`--dump` on the GPU box:   python tools/outlier_f64.py [--dump out.npy | --hip out.npy] [model] [dtype] [outliers 0/1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd import calmfile as cf
from calm_amd.host import HostModel

args = sys.argv[1:]
dump = hipf = None
if args and args[0] == "--dump":
    dump, args = args[1], args[2:]
elif args and args[0] == "--hip":
    hipf, args = args[1], args[2:]
name = args[0] if len(args) > 0 else "mistral-7b"
dtype = args[1] if len(args) > 1 else "fp8"
outl = bool(int(args[2])) if len(args) > 2 else True
spec = cf.SPECS[name]
TOKS = [11, 12, 13, 14, 15, 16]
tensors, md = cf.synth_model_big(spec, dtype, 5, outliers=outl)
model = HostModel(tensors, md, context=64)
if dump:
    from calm_amd.host import HipBackend

    be = HipBackend(model)
    np.save(dump, np.stack([be.forward(t, 0, 0).copy() for t in TOKS]))
    be.close()
    sys.exit(0)

from oracle import oracle

from oracle.f64_step import position0_logits_f64

ref = oracle.RefBackend(model) if oracle.have_ref() else oracle.OracleBackend(model)
orc = oracle.OracleBackend(model)
hip = np.load(hipf) if hipf else None
L64 = position0_logits_f64(tensors, md, spec, dtype, TOKS)
print(f"{name} {dtype} outliers={outl}: position 0 of {len(TOKS)} first tokens; max|d| / max|logit| against a float64 evaluation of the same step")
for i, t in enumerate(TOKS):
    lr, lo = ref.forward(t, 0, 0).copy(), orc.forward(t, 0, 0).copy()
    err = lambda a: float(np.abs(a.astype(np.float64) - L64[i]).max() / np.abs(L64[i]).max())
    print(f"  token {t}: max|logit| {np.abs(L64[i]).max():6.2f}   CPU reference {err(lr):.2e}   restatement {err(lo):.2e}" +
          (f"   HIP {err(hip[i]):.2e}   |   HIP vs reference {float(np.abs(hip[i] - lr).max() / np.abs(lr).max()):.2e}" if hip is not None else ""), flush=True)
ref.close()
orc.close()
