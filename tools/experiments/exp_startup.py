#!/usr/bin/env python3
"""Start-up probe (tools/experiments/exp_mega.hip: k_startup): what delays a streaming kernel's first tile?  Run on the GPU box."""
import ctypes as C, os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_mega.so")
src = os.path.join(here, "exp_mega.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
lib = C.CDLL(so)
lib.exp_startup.argtypes = [C.c_int, C.c_int]
for grid in (256, 512):
    for mode in (0, 1, 2):
        lib.exp_startup(mode, grid)
