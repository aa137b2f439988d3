#!/usr/bin/env python3
"""What this box's memory system delivers to a pure 16-byte-per-lane streaming read:
HBM (2 GiB buffer) and Infinity-Cache-resident (64 MiB buffer), plain vs non-temporal loads."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calm_amd.host import load_lib

lib = load_lib()
for label, nbytes, iters in (("HBM 2GiB", 2 << 30, 10), ("HBM 512MiB", 512 << 20, 20), ("MALL 128MiB", 128 << 20, 50), ("MALL 64MiB", 64 << 20, 100), ("L2 16MiB", 16 << 20, 200)):
    for nt in (0, 1):
        print(f"{label:12s} nt={nt}: {lib.calm_hip_membench(nbytes, nt, iters):8.1f} GB/s", flush=True)
