#!/usr/bin/env python3
"""Driver of the kernel-overlap experiment (tools/experiments/exp_overlap.hip); run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp_overlap.so")
src = os.path.join(here, "exp_overlap.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = C.CDLL(so)
lib.exp_concurrency.restype = C.c_int
lib.exp_chain.restype = C.c_double
lib.exp_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
lib.exp_concurrency_anyorder.restype = C.c_int
lib.exp_set_depth.argtypes = [C.c_int]
L = 8
if "deep" in sys.argv:
    # round-2 opener (exp_overlap.hip: k_deep): one persistent launch, coordinator wave + 7 streamer waves per CU holding
    # DEPTH x 8 KiB each in registers across phase boundaries, against the launch-per-kernel chain.  The checksums must match.
    lib.exp_deep.restype = C.c_double
    lib.exp_deep.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.exp_set_real.argtypes = [C.c_int]
    for real in (0, 1):  # 0: tiles are only touched (pure streaming); 1: every tile goes through the fp8 decode inner loop
        lib.exp_set_real(real)
        what = "fp8 decode loop per tile" if real else "tiles only touched"
        cs = C.c_double(0)
        us = lib.exp_chain(0, 1, L, 10, C.byref(cs), 256)
        print(f"[{what}] launch per kernel (graph)         : {us:8.2f} us/layer  ({218.1/us:6.2f} TB/s)  checksum {cs.value:.6f}", flush=True)
        for depth in ((2, 4, 5) if real else (2, 4, 6)):
            cs = C.c_double(0)
            us = lib.exp_deep(depth, L, 10, C.byref(cs))
            print(f"[{what}] persistent, {depth} x 8 KiB per streamer : {us:8.2f} us/layer  ({218.1/us:6.2f} TB/s)  checksum {cs.value:.6f}", flush=True)
    lib.exp_set_real(0)
    sys.exit(0)
if "anyorder" in sys.argv:
    # same-stream overlap through hipExtAnyOrderLaunch (exp_overlap.hip modes 3 / 4); eager launches only
    seen = lib.exp_concurrency_anyorder(256)
    rows = [(0, 1, 0), (0, 0, 0)]
    if seen > 0:
        rows += [(3, 0, 0), (3, 0, 2), (3, 0, 4), (4, 0, 0), (4, 0, 2), (3, 1, 0)]
    label = {0: "plain", 3: "any-order chain (acquire fence)", 4: "any-order chain (system-scope)"}
    for mode, graph, depth in rows:
        lib.exp_set_depth(depth)
        cs = C.c_double(0)
        us = lib.exp_chain(mode, graph, L, 10, C.byref(cs), 256)
        print(f"grid=256 graph={graph} depth={depth} mode={label[mode]:34s}: {us:8.2f} us/layer  ({218.1/us:6.2f} TB/s)  checksum {cs.value:.6f}", flush=True)
    sys.exit(0)
lib.exp_concurrency(256)
lib.exp_persist.restype = C.c_double
lib.exp_persist.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
for grid in (256, 512):
    for variant in (0, 1):
        cs = C.c_double(0)
        us = lib.exp_persist(variant, L, 6, C.byref(cs), grid)
        print(f"grid={grid} persistent variant={'acquire-fence' if variant == 0 else 'system-scope'}: {us:8.2f} us/layer  ({218.1/us:6.2f} TB/s)  checksum {cs.value:.6f}", flush=True)
names = {0: "plain", 1: "chained(acquire fence)", 2: "chained(system-scope loads)"}
for grid in (256,):
    for graph, mode in ((1, 0),):
        cs = C.c_double(0)
        us = lib.exp_chain(mode, graph, L, 6, C.byref(cs), grid)
        print(f"grid={grid} graph={graph} mode={names[mode]:28s}: {us:8.2f} us/layer  ({218.1/us:6.2f} TB/s)  checksum {cs.value:.6f}", flush=True)
