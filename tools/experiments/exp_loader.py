#!/usr/bin/env python3
"""Loader study for the engine gate (tools/experiments/exp_engine2.hip: k_loader_only): the rate at which LDS-DMA alone streams a layer's
weights -- loader waves per CU, fills in flight per wave, slot placement, cache policy -- against the same stream through
registers.  One persistent launch over 8 layers of the chain's four phases; no consumers, no edges."""
import ctypes as C
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_engine2.so")
src = os.path.join(here, "exp_engine2.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = C.CDLL(so)
lib.exp_loader_only.restype = C.c_double
lib.exp_loader_only.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
L = 8
for cfg, label in ((900, "registers, 256 x 8 waves, 16 KiB per wave"), (901, "registers, 512 x 4 waves")):
    us = lib.exp_loader_only(cfg, 0, L, 10)
    print(f"{label:48s}: {us:7.2f} us/layer ({218.1/us:5.2f} TB/s)", flush=True)
for cfg in (121, 131, 141, 130, 221, 231, 241, 230, 331, 421, 411):
    nl, d, nt = cfg // 100, cfg // 10 % 10, cfg % 10
    row = []
    for run in (0, 1, 4, 16):
        us = lib.exp_loader_only(cfg, run, L, 10)
        row.append(f"run {run:2d}: {us:6.2f} us ({218.1/us:4.2f} TB/s)")
    print(f"LDS-DMA {nl} loader wave(s) x {d} fills of 16 KiB in flight, {'nt' if nt else 'default policy'}: " + " | ".join(row), flush=True)
