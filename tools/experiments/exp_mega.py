#!/usr/bin/env python3
"""Driver of the second persistent-launch experiment (tools/experiments/exp_mega.hip); run on the GPU box.
usage: exp_mega.py [dbg] [cfg ...]   cfg = NS*1000 + NH*100 + DEPTH*10 + SYS"""
import ctypes as C
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_mega.so")
src = os.path.join(here, "exp_mega.hip")
dep = os.path.join(here, "exp_overlap.hip")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(dep)):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = C.CDLL(so)
lib.exp_chain.restype = C.c_double
lib.exp_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
lib.exp_mega.restype = C.c_double
lib.exp_mega.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
lib.exp_set_real.argtypes = [C.c_int]
L = 8
dbg = "dbg" in sys.argv
cfgs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4140, 4160, 4440, 4460, 4441, 4461, 4260, 4420, 7140, 6240]
for real in (0, 1):
    lib.exp_set_real(real)
    what = "fp8 loop" if real else "touch   "
    cs = C.c_double(0)
    us = lib.exp_chain(0, 1, L, 10, C.byref(cs), 256)
    print(f"[{what}] launch per kernel (graph)      : {us:8.2f} us/layer ({218.1/us:5.2f} TB/s) checksum {cs.value:.6f}", flush=True)
    for cfg in cfgs:
        cs = C.c_double(0)
        us = lib.exp_mega(cfg, real | (2 if dbg else 0), L, 10, C.byref(cs))
        ns, nh, d, sy = cfg // 1000, cfg // 100 % 10, cfg // 10 % 10, cfg % 10
        print(f"[{what}] mega NS={ns} NH={nh} DEPTH={d} {'sys' if sy else 'sc1'}     : {us:8.2f} us/layer ({218.1/us:5.2f} TB/s) checksum {cs.value:.6f}", flush=True)
