"""Reproduction of the round-4 red GPU suite (GPUTEST_r04: test_matvec_is_linear_at_full_size, second call all zeros).

    python tools/experiments/race_repro.py TESTLIB [ITERS]

Loads a test-hook library (tools/experiments/libcalm_hip_test_r04.so = round 4's build, whose calm_hip_test_matvec zeroes its
output with a NULL-stream hipMemset ahead of a launch on the non-blocking decode stream; calm_amd/libcalm_hip_test.so = this
round's, every fill ordered on the decode stream) and hammers the failing shape -- a 256 x 14336 fp8 matvec (k_attn_out:
x += W.v on a zeroed x) -- alternating the input between x and 8 x, the test's own property.  Counts calls whose output is
all zeros / differs from the first answer.  Output: one JSON line.  Test infrastructure, not product code.
(Round 5, one MI355X: 0 of 3000 calls with either library -- the driver's red run did not reproduce in isolation; profiles/r05_gpu_tests.txt.)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    path = os.path.abspath(sys.argv[1])
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    lib = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    lib.calm_hip_test_matvec.restype = None
    lib.calm_hip_test_matvec.argtypes = [C.c_int, C.c_void_p, fp, fp, C.c_int, C.c_int]
    n, d = 14336, 256
    rng = np.random.default_rng(1)
    # every e5m2 code that is finite: exponent field below 31
    codes = np.array([c for c in range(256) if (c >> 2) & 31 != 31], dtype=np.uint8)
    w = codes[rng.integers(0, len(codes), size=(d, n))].copy()
    x = rng.standard_normal(n).astype(np.float32)
    x8 = x * np.float32(8)
    out = np.empty(d, dtype=np.float32)

    def call(v):
        out.fill(np.float32(-7.0))  # poison: a hook that copies nothing back is seen as such
        lib.calm_hip_test_matvec(8, w.ctypes.data, v.ctypes.data_as(fp), out.ctypes.data_as(fp), n, d)
        return out.copy()

    ref = call(x)
    assert np.abs(ref).max() > 0, "the very first call returned zeros"
    ref8 = ref * np.float32(8)

    zeros = wrong = 0
    t0 = time.time()
    for i in range(iters):
        v, r = (x8, ref8) if i & 1 else (x, ref)
        o = call(v)
        if not o.any():
            zeros += 1
        elif not np.array_equal(o, r):
            wrong += 1
    print(json.dumps({"lib": os.path.relpath(path, ROOT), "calls": iters, "all_zero": zeros, "other_mismatch": wrong, "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
