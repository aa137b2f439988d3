#!/usr/bin/env python3
"""Is a matvec kernel's tail -- which workgroups exit late -- the same from launch to launch?  Per-wave exit stamps (the -DCALM_TIMELINE
build of tools/timeline.py) of the LAST launch of perf_stage_hip's sweep (= the last layer's weights), taken REPS times in one process;
prints the correlation of the workgroups' mean exit times between repetitions, and between two DIFFERENT layer counts (other weights
behind the same workgroup indices).  A persistent pattern could be dealt against statically (per process); a fresh one cannot.

    python tools/tail_persistence.py [model] [dtype] [layers] [stage]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libcalm_hip_tl.so")
assert os.path.exists(SO), "build it first: python tools/timeline.py --build-only"
os.environ["CALM_HIP_LIB"] = SO
import numpy as np  # noqa: E402

from calm_amd import calmfile as cf  # noqa: E402
from calm_amd.host import STAGES, HipBackend, HostModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mistral-7b"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
stage = sys.argv[4] if len(sys.argv) > 4 else "ffn_up"
REPS = 6
waves = 8192


def sample(L):
    spec = cf.SPECS[name]
    model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
    b = HipBackend(model, device_synth=(spec, dtype, 1, L))
    lib = b.lib._product
    lib.calm_tl_arm.argtypes = [C.c_int]
    lib.calm_tl_read.argtypes = [C.c_void_p, C.c_int]
    for pos in range(32):
        b.forward(5 + pos, pos, 0)
    i = STAGES.index(stage)
    wpb = 8 if stage == "ffn_down" else 4
    out = []
    for _ in range(REPS):
        lib.calm_tl_arm(waves)
        b.stage_us(i, 2)
        buf = np.zeros((waves, 8), dtype=np.uint64)
        lib.calm_tl_read(buf.ctypes.data, waves)
        t = buf.astype(np.int64) * 10
        live = t[:, 3] > 0
        t0 = t[live][:, 0].min()
        ex = np.where(live, (t[:, 3] - t0) / 1e3, np.nan)
        nb = int(np.nonzero(live)[0].max()) // wpb + 1
        out.append(np.nanmean(ex[: nb * wpb].reshape(nb, wpb), axis=1))
    b.close()
    return np.array(out)


a = sample(int(sys.argv[3]) if len(sys.argv) > 3 else 8)
print(f"{name} {dtype} {stage}: {a.shape[1]} workgroups, {REPS} launches of the last layer; workgroup mean exit us: mean {a.mean():.2f}, "
      f"spread within a launch (std) {a.std(axis=1).mean():.2f}, of one workgroup across launches (std) {a.std(axis=0).mean():.2f}")
c = np.corrcoef(a)
print("correlation of the workgroups' exit times between launches (same weights):")
print(np.array2string(c, precision=2, suppress_small=True))
xcd = np.array([[r[(np.arange(len(r)) % 8) == x].mean() for x in range(8)] for r in a])
print("per-XCD mean exit by launch:\n" + np.array2string(xcd, precision=2))
b2 = sample(4)
print(f"another model instance (4 layers: other weights behind the same workgroup indices): correlation with the first, launch 0 vs 0: "
      f"{np.corrcoef(a[0], b2[0])[0, 1]:.2f}; per-XCD means {np.array2string(np.array([b2[0][(np.arange(len(b2[0])) % 8) == x].mean() for x in range(8)]), precision=2)}")
