#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of every kernel in libcalm_hip.so whose demangled name contains one of the given substrings (no GPU):
    python tools/kernel_regs.py k_attn_vt 'k_ffn_up<8'"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.environ.get("CALM_HIP_LIB", os.path.join(ROOT, "calm_amd", "libcalm_hip.so"))
with tempfile.TemporaryDirectory() as d:
    so = os.path.join(d, "lib.so")
    subprocess.run(["cp", lib, so], check=True)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], cwd=d, check=True, capture_output=True)
    obj = [f for f in os.listdir(d) if "amdgcn" in f][0]
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(d, obj)], capture_output=True, text=True, check=True).stdout
blocks = notes.split("  - .agpr_count:")[1:]
rows = []
for b in blocks:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    f = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1))
    rows.append((name, f("vgpr_count"), f("sgpr_count"), f("group_segment_fixed_size"), f("private_segment_fixed_size")))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (name, v, s, lds, scr), dn in zip(rows, dem):
    short = re.sub(r"\(.*", "", dn.replace("void calm::", ""))
    if not sys.argv[1:] or any(a in short for a in sys.argv[1:]):
        print(f"{v:4d} vgpr {s:4d} sgpr {lds:6d} lds {scr:4d} scratch  {short}")
