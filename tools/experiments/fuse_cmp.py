import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel, load_lib
lib = load_lib()
spec = cf.SPECS["mistral-7b"]; L = 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
model = HostModel(cf.stub_tensors(spec, "fp8", L), cf.dataclasses.replace(spec, n_layers=L).metadata("fp8"))
outs = {}
for v in (0, 1):
    lib.calm_hip_configure(b"qkv_attn", v)
    be = HipBackend(model, device_synth=(spec, "fp8", 1, L))
    o = []
    for pos in range(N):
        o.append(be.forward(5 + pos, pos, 0).copy())
    outs[v] = np.stack(o); be.close()
for pos in range(N):
    a, b = outs[0][pos], outs[1][pos]
    print(pos, float(np.abs(a - b).max() / np.abs(a).max()), bool(np.isfinite(b).all()))
print("timeouts", lib.calm_hip_query(b"fuse_timeouts", 0))
