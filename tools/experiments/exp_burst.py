#!/usr/bin/env python3
"""One-shot deep-prefetch probe (tools/experiments/exp_mega.hip: k_burst); run on the GPU box."""
import ctypes as C, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_mega.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_mega.hip")], check=True)
lib = C.CDLL(so)
lib.exp_burst.argtypes = [C.c_int, C.c_int]
for ns in (4, 8):
    for depth in (2, 4, 6, 12):
        if ns * depth <= 48:
            lib.exp_burst(depth, ns)
