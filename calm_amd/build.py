"""Build recipes (explicit hipcc / gcc invocations, in-tree outputs).

    python -m calm_amd.build            # libcalm_hip.so
    python -m calm_amd.build --all      # + the oracle checker and, when /root/reference exists, oracle/_ref

The product is calm_amd/libcalm_hip.so: hand-written HIP for gfx950 behind the C ABI of
include/calm_hip.h.  hipcc cross-compiles without a GPU.  The built .so is git-ignored but travels to
the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "calm_amd", "csrc")
LIB_HIP = os.path.join(ROOT, "calm_amd", "libcalm_hip.so")
LIB_HIP_TEST = os.path.join(ROOT, "calm_amd", "libcalm_hip_test.so")  # unit-test hooks: tests/ and tools/ only
ORACLE_DIR = os.path.join(ROOT, "oracle")
REFERENCE = os.environ.get("CALM_REFERENCE", "/root/reference")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-kernarg-preload-count: the dispatcher hands a wave the first dwords of its kernel arguments in SGPRs (gfx940+) instead of
# the wave loading them -- one memory round trip less at the head of every launch; kernels.hip.h orders its arguments for it
PRELOAD_FLAGS = ["-mllvm", "-amdgpu-kernarg-preload-count=14"]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
HIP_FLAGS = BASE_FLAGS + PRELOAD_FLAGS


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def hip_sources():
    inc = os.path.join(ROOT, "include")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(inc, f) for f in sorted(os.listdir(inc))]


def csrc_sha() -> str:
    """hash of the kernel and ABI sources: stamps a profile so that a stale one is recognised (no .git on the GPU box)"""
    import hashlib

    h = hashlib.sha256()
    for f in hip_sources():
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_hip(force: bool = False) -> str:
    """compile every HIP translation unit for gfx950 into libcalm_hip.so"""
    srcs = hip_sources()
    jobs = []
    if force or not _newer(LIB_HIP, srcs):
        jobs.append([HIPCC, *HIP_FLAGS, "-shared", "-o", LIB_HIP, os.path.join(CSRC, "infer_hip.hip")])
    if force or not _newer(LIB_HIP_TEST, srcs):
        jobs.append([HIPCC, *HIP_FLAGS, "-shared", "-o", LIB_HIP_TEST, os.path.join(CSRC, "test_hooks.hip")])
    # the two libraries are independent translation units (the second includes the first's source): compile side by side
    procs = []
    for cmd in jobs:
        print("+", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return LIB_HIP


def build_variant(tag: str, extra=(), base=None) -> str:
    """an A/B build of the product library for tools/ (calm_amd/libcalm_hip_<tag>.so, loaded through CALM_HIP_LIB): the same
    source with other flags / -D switches"""
    out = os.path.join(ROOT, "calm_amd", f"libcalm_hip_{tag}.so")
    if not _newer(out, hip_sources() + [os.path.abspath(__file__)]):
        _run([HIPCC, *(BASE_FLAGS if base is None else base), *extra, "-shared", "-o", out, os.path.join(CSRC, "infer_hip.hip")])
    return out


def build_oracle(force: bool = False) -> str:
    """the CPU checker (test infrastructure): our C restatement, and the real reference when present"""
    args = ["make", "-C", ORACLE_DIR]
    if force:
        args.append("-B")
    _run(args + ["liboracle.so"])
    if os.path.isdir(os.path.join(REFERENCE, "src")):
        _run(args + [f"REF={REFERENCE}", "ref"])
        if os.path.exists(LIB_HIP):
            _run(args + [f"REF={REFERENCE}", "_ref/run_hip"])
    return os.path.join(ORACLE_DIR, "liboracle.so")


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    if "--all" in sys.argv:
        build_oracle(force="--force" in sys.argv)
