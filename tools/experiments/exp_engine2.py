#!/usr/bin/env python3
"""Driver of the second-cut LDS-DMA loader/consumer engine (tools/experiments/exp_engine2.hip) -- round 3's time-boxed gate; run on the GPU box.
usage: exp_engine2.py [cfg ...] [noedge]     cfg = R * 10 + D (ring slots of 16 KiB, fills in flight)
Gate: <= 42 us/layer against ~47 for launch-per-kernel on this chain."""
import ctypes as C
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_engine2.so")
src = os.path.join(here, "exp_engine2.hip")
dep = os.path.join(here, "exp_overlap.hip")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(dep)):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = C.CDLL(so)
lib.exp_chain.restype = C.c_double
lib.exp_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
lib.exp_set_real.argtypes = [C.c_int]
lib.exp_engine2_ref.restype = C.c_double
lib.exp_engine2_ref.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
lib.exp_engine2.restype = C.c_double
lib.exp_engine2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
lib.exp_engine2_knobs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
L = 8
cfgs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [83, 84, 82]
for real in (0, 1):
    what = "fp8 loop" if real else "touch   "
    lib.exp_set_real(real)
    cs = C.c_double(0)
    us = lib.exp_chain(0, 1, L, 10, C.byref(cs), 256)
    print(f"[{what}] launch per kernel (exp_chain, graph)                : {us:8.2f} us/layer ({218.1/us:5.2f} TB/s)", flush=True)
    us = lib.exp_engine2_ref(real, L, 10, C.byref(cs))
    print(f"[{what}] launch per phase over the granule buffers (checker) : {us:8.2f} us/layer   checksum {cs.value:.6f}", flush=True)
    for noedge, thin, tree in ((0, 1, 0), (0, 1, 1), (0, 0, 1)) + (((1, 1, 0),) if "noedge" in sys.argv else ()):
        lib.exp_engine2_knobs(noedge, thin, 0 if noedge else 1, tree)
        for cfg in cfgs:
            cs, bad = C.c_double(0), C.c_int(0)
            us = lib.exp_engine2(cfg, real, L, 10, C.byref(cs), C.byref(bad))
            tagl = "NO EDGES (diagnostic, wrong results)" if noedge else f"values differing from the checker: {bad.value}"
            print(f"[{what}] engine ring {cfg // 10} x 16 KiB, {cfg % 10} fills in flight, thinning {'on ' if thin else 'off'}, {'two-level' if tree else 'flat     '} gather: "
                  f"{us:8.2f} us/layer ({218.1/us if us > 0 else 0:5.2f} TB/s)   {tagl}", flush=True)
lib.exp_engine2_knobs(0, 1, 0, 0)
