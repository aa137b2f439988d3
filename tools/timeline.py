#!/usr/bin/env python3
"""Where does a decode kernel's time go?  Builds the library a second time with -DCALM_TIMELINE (per-wave 100 MHz wall-clock
stamps inside the row engine: entry / LDS image built / first tile done / exit), launches each matvec stage of a BASELINE shape
back to back over the layers (perf_stage_hip) and reads the stamps of the LAST launch: when its waves start, how long the
prologue takes, how the exits are distributed (the tail).  Run on the GPU box:

    python tools/timeline.py [model] [dtype] [layers]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PRELOAD = "--no-preload" not in sys.argv  # kernel-argument preloading, as the product is built (calm_amd/build.py)
DEFS = os.environ.get("TL_DEFS", "").split()  # further -D switches of an A/B variant (TL_TAG names its library)
TAG = os.environ.get("TL_TAG", "")
SO = os.path.join(ROOT, "tools", f"libcalm_hip_tl{'_' + TAG if TAG else ''}{'' if PRELOAD else '_nopreload'}.so")
src = os.path.join(ROOT, "calm_amd", "csrc", "infer_hip.hip")
deps = [os.path.join(ROOT, "calm_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "calm_amd", "csrc"))]
if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DCALM_TIMELINE", *DEFS, *(["-mllvm", "-amdgpu-kernarg-preload-count=14"] if PRELOAD else []),
                    "-shared", "-o", SO, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["CALM_HIP_LIB"] = SO

import numpy as np  # noqa: E402

from calm_amd import calmfile as cf  # noqa: E402
from calm_amd.host import STAGES, HipBackend, HostModel  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if len(args) > 0 else "mistral-7b"
dtype = args[1] if len(args) > 1 else "fp8"
L = int(args[2]) if len(args) > 2 else 8
spec = cf.SPECS.get(name) or cf.ARCH_SPECS[name]
model = HostModel(cf.stub_tensors(spec, dtype, L), cf.dataclasses.replace(spec, n_layers=L).metadata(dtype))
b = HipBackend(model, device_synth=(spec, dtype, 1, L))
lib = b.lib._product
for kv in os.environ.get("TL_KNOBS", "").split():  # e.g. TL_KNOBS="qkv_attn=0 fuse_dbg=1"
    k, v = kv.split("=")
    lib.calm_hip_configure(k.encode(), int(v))
lib.calm_tl_arm.argtypes = [C.c_int]
lib.calm_tl_read.argtypes = [C.c_void_p, C.c_int]
for pos in range(int(os.environ.get("TL_POS", "64"))):  # the cache length the attention stage then meets
    b.forward(5 + pos, pos, 0)
ONLY = os.environ.get("TL_ONLY", "").split()  # e.g. TL_ONLY=attn
waves = 8192
print(f"{name} {dtype} L={L}; us after the launch's first wave started; stage time = perf_stage_hip of the INSTRUMENTED build")
print(f"{'kernel':9s} {'us/launch':>9s} {'waves':>6s} | last wave start | image built p50 / max | first tile done p50 / max | exits p1 / p10 / p50 / p90 / p99 / last")
for i, st in enumerate(STAGES):
    if ONLY and st not in ONLY:
        continue
    if st == "attn":
        # k_attn (contexts up to 384 positions): entry / cache length known / positions folded in / lane groups merged / barrier / exit
        lib.calm_tl_arm(waves)
        us, _ = b.stage_us(i, 2)
        buf = np.zeros((waves, 8), dtype=np.uint64)
        lib.calm_tl_read(buf.ctypes.data, waves)
        t = buf.astype(np.int64) * 10
        a = t[t[:, 3] > 0]
        t0 = a[:, 0].min()
        col = lambda k: (a[:, k] - t0) / 1e3
        med = lambda k: float(np.median(col(k)))
        print(f"{st:9s} {us:9.2f} {len(a):6d} | last wave start {col(0).max():5.2f} | p50: cache length known {med(1):5.2f}, positions folded in {med(2):5.2f}, lane groups merged "
              f"{med(4):5.2f}, past the barrier {med(5):5.2f}, exit {med(3):5.2f} (last {col(3).max():5.2f})")
        continue
    lib.calm_tl_arm(waves)
    us, _ = b.stage_us(i, 2)
    buf = np.zeros((waves, 8), dtype=np.uint64)
    lib.calm_tl_read(buf.ctypes.data, waves)
    t = buf.astype(np.int64) * 10
    if st == "qkv" and lib.calm_hip_configure(b"qkv_attn", -1) >= 1:
        # k_qkv_attn: the first n_heads workgroups are the attention role -- entry / q here / old positions folded in, past the barrier /
        # this token's k, v rows here / exit; the row engine's stamps follow below
        na = spec.n_heads * 4
        at = t[:na][t[:na, 3] > 0]
        t0 = t[t[:, 3] > 0][:, 0].min()
        if len(at):
            c = lambda k: (at[:, k] - t0) / 1e3
            f = lambda k: f"{float(np.median(c(k))):5.2f} / {c(k).max():5.2f}"
            print(f"{'qkv:attn':9s} {'':9s} {len(at):6d} | entry {f(0)} | q here {f(1)} | own rows folded in {f(5)} | past the barrier {f(2)} | k, v here {f(4)} | exit {f(3)}   (p50 / max)")
            if os.environ.get("TL_HEADS"):
                for h in range(0, spec.n_heads):
                    w = (t[4 * h:4 * h + 4] - t0) / 1e3
                    print(f"   wg {h:2d}: q here " + " ".join(f"{x:5.2f}" for x in w[:, 1]) + " | folded " + " ".join(f"{x:5.2f}" for x in w[:, 5]) + f" | barrier {w[0, 2]:5.2f} | kv {w[0, 4]:5.2f} | exit {w[0, 3]:5.2f}")
        t[:na] = 0
        wo = t[t[:, 7] > 0]
        if len(wo):  # k_qkv_attn WO: the row engine's waves go on to the output projection
            c = lambda k: (wo[:, k] - t0) / 1e3
            f = lambda k: f"{np.percentile(c(k), 10):5.2f} / {float(np.median(c(k))):5.2f} / {c(k).max():5.2f}"
            print(f"{'qkv:wo':9s} {'':9s} {len(wo):6d} | rows of wo asked for {f(4)} | every flag up {f(5)} | image built {f(6)} | exit {f(7)}   (p10 / p50 / max)")
    a = t[t[:, 3] > 0]
    t0 = a[:, 0].min()
    ex = (a[:, 3] - t0) / 1e3
    ft = (a[:, 2][a[:, 2] > 0] - t0) / 1e3
    im = (a[:, 1] - t0) / 1e3
    pc = lambda v, q: float(np.percentile(v, q))
    print(f"{st:9s} {us:9.2f} {len(a):6d} | {(a[:, 0].max() - t0) / 1e3:6.2f} | {pc(im, 50):6.2f} {im.max():6.2f} | {pc(ft, 50):6.2f} {ft.max():6.2f} | "
          f"{pc(ex, 1):6.2f} {pc(ex, 10):6.2f} {pc(ex, 50):6.2f} {pc(ex, 90):6.2f} {pc(ex, 99):6.2f} {ex.max():6.2f}")
    if st in ("ffn_up", "ffn_down"):
        wpb = 4 if st == "ffn_up" else 8
        idx = np.nonzero(t[:, 3] > 0)[0]
        blk = idx // wpb
        print("          exits by XCD (block % 8), p50 / max: " + "  ".join(f"{x}: {pc(ex[blk % 8 == x], 50):5.1f}/{ex[blk % 8 == x].max():5.1f}" for x in range(8)))
        # who is late?  by wave slot within the workgroup, by workgroup half (b < 256: first resident round), and how much of
        # the spread is BETWEEN workgroups (workgroup means) vs WITHIN them
        wv = idx % wpb
        print("          exit p50 by wave slot in the workgroup: " + " ".join(f"{w}:{pc(ex[wv == w], 50):5.1f}" for w in range(wpb))
              + "   by workgroup index range: " + " ".join(f"[{lo},{lo + 128}):{pc(ex[(blk >= lo) & (blk < lo + 128)], 50):5.1f}" for lo in range(0, int(blk.max()) + 1, 128)))
        nb = int(blk.max()) + 1
        wg_mean = np.array([ex[blk == g].mean() for g in range(nb)])
        wg_rng = np.array([ex[blk == g].max() - ex[blk == g].min() for g in range(nb)])
        print(f"          workgroup mean exit: min {wg_mean.min():.1f} p50 {np.median(wg_mean):.1f} max {wg_mean.max():.1f};  spread inside a workgroup (max - min): p50 {np.median(wg_rng):.2f} max {wg_rng.max():.2f}")
        hist, edges = np.histogram(ex, bins=10)
        print("          exit histogram: " + " ".join(f"{edges[k]:.1f}-{edges[k + 1]:.1f}:{hist[k]}" for k in range(len(hist))))
b.close()
