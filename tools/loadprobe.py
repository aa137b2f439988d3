import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.getcwd())
import numpy as np
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel, load_lib
t1 = time.perf_counter(); print(f"imports {t1-t0:.2f}s", flush=True)
lib = load_lib(); t2 = time.perf_counter(); print(f"dlopen {t2-t1:.2f}s", flush=True)
lib.init_hip(); t3 = time.perf_counter(); print(f"init_hip {t3-t2:.2f}s", flush=True)
spec = cf.SPECS["mistral-7b"]
tensors, md = cf.synth_model_big(spec, "fp8", 1); t4 = time.perf_counter(); print(f"host synth 7 GB {t4-t3:.2f}s", flush=True)
model = HostModel(tensors, md)
import ctypes as C
# upload only
t5 = time.perf_counter()
ptrs = []
for n, a in tensors.items():
    if n.startswith("model."):
        a = np.ascontiguousarray(a); ptrs.append(lib.upload_hip(a.ctypes.data, a.nbytes))
t6 = time.perf_counter(); print(f"upload_hip x {len(ptrs)}: {t6-t5:.2f}s", flush=True)
for p in ptrs: lib.free_hip(p)
t7 = time.perf_counter(); print(f"free {t7-t6:.2f}s", flush=True)
be = HipBackend(model); t8 = time.perf_counter(); print(f"HipBackend (upload + prepare) {t8-t7:.2f}s", flush=True)
be.forward(17, 0, 0); t9 = time.perf_counter(); print(f"first forward (graph capture) {t9-t8:.2f}s", flush=True)
be.close()
