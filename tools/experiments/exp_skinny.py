#!/usr/bin/env python3
"""T tokens through one weight stream (tools/experiments/exp_skinny.hip): a Mistral-7B fp8 layer's matrix shapes, 4 / 8 tokens at once on the
v_mfma_f32_4x4x4_16b_f16 path, against the decode kernels' time for ONE token (tools/tune.py) -- is a multi-token row engine worth
building for chunks of 3-16 prompt tokens?"""
import ctypes as C
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_skinny.so")
src = os.path.join(here, "exp_skinny.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = C.CDLL(so)
lib.exp_skinny.restype = C.c_double
lib.exp_skinny.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for name, M, K, one in (("qkv", 6144, 4096, 7.7), ("wo", 4096, 4096, 5.2), ("ffn-up (w1, w3)", 28672, 4096, 20.6)):
    for T in (4, 8):
        err = C.c_double()
        us = lib.exp_skinny(T, M, K, 8, 20, C.byref(err))
        mb = M * K / 1e6
        print(f"{name:16s} {M:6d} x {K}: T = {T}: {us:7.2f} us per launch = {mb / us * 1e3:6.0f} GB/s of weights ({us / T:6.2f} us per token; one decode token {one} us); "
              f"error vs float64 {err.value:.2e}", flush=True)
