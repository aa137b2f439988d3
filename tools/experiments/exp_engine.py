#!/usr/bin/env python3
"""Driver of the LDS-DMA loader/consumer engine experiment (tools/experiments/exp_engine.hip); run on the GPU box.
usage: exp_engine.py [cfg ...]   cfg = NL*100000 + NC*1000 + R*10 + D"""
import ctypes as C
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_engine.so")
src = os.path.join(here, "exp_engine.hip")
dep = os.path.join(here, "exp_overlap.hip")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(dep)):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = C.CDLL(so)
lib.exp_chain.restype = C.c_double
lib.exp_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
lib.exp_engine.restype = C.c_double
lib.exp_engine.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
lib.exp_set_real.argtypes = [C.c_int]
L = 8
cfgs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [104166, 107166, 104084, 204164, 206164, 206163, 404162]
lib.exp_engine_noedge.argtypes = [C.c_int]
if "noedge" in sys.argv:
    # diagnostic: the grid-wide wait skipped (results wrong): what the engine streams at when no edge ever holds it up
    lib.exp_engine_noedge(1)
for real in (0, 1):
    lib.exp_set_real(real)
    what = "fp8 loop" if real else "touch   "
    cs = C.c_double(0)
    us = lib.exp_chain(0, 1, L, 10, C.byref(cs), 256)
    print(f"[{what}] launch per kernel (graph)        : {us:8.2f} us/layer ({218.1/us:5.2f} TB/s) checksum {cs.value:.6f}", flush=True)
    for cfg in cfgs:
        for sysv in (0,):
            cs = C.c_double(0)
            us = lib.exp_engine(cfg, real | (2 if sysv else 0), L, 10, C.byref(cs))
            nl, nc, r, d = cfg // 100000, cfg // 1000 % 100, cfg // 10 % 100, cfg % 10
            print(f"[{what}] engine {nl} loader(s) + {nc} consumers, ring {r} x 8 KiB, {d} tiles in flight: {us:8.2f} us/layer ({218.1/us:5.2f} TB/s) checksum {cs.value:.6f}", flush=True)
