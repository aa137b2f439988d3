"""pytest configuration.

Markers:
    gpu -- needs a real MI355X; these are the parity tests proper and call through the C ABI.
           Everything else runs on CPU (`pytest -m "not gpu"`).

oracle/ is imported here and in the tests only (it is test infrastructure, never product code).
"""
import os
import sys

import numpy as np
import pytest

# the CPU checkers (oracle/liboracle.so, oracle/_ref) are OpenMP code with tiny parallel regions: on a many-core
# box the default team size makes every region a scheduling storm (a 150-token tiny-model run took 290 s)
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"

GOLDEN_CASES = ["tiny_fp16", "tiny_fp8", "tiny_gf4", "moe_fp8", "ln_gelu_clip_fp16", "par_fp8", "bias_tied_gf4", "sink_fp16", "ragged_fp8",
                "partial_rope_fp16", "dbrx_like_fp8", "mqa_hd96_fp16", "moe_gf4", "hd256_sink_fp8", "moe6_fp8", "moe12_ln_fp16",
                "fuse_hd64_bias_fp16", "fuse_hd128_sink_fp16"]


# ---- tolerances (max |delta| over a token's logits / max |logit|) ----------------------------------------------
# LOGIT_TOL: whole decode steps with the fp16 KV cache -- the north star's 1e-3, for fp16 AND fp8 / gf4 weights (weights decode exactly,
#   activations stay fp32).  Measured: <= 3.6e-4 at 32 layers (profiles/r05_fp8kv.txt, kvbits 16 rows).
# FP8KV_TOL: the same comparison with an fp8 (e5m2) KV cache on both sides, models of AT MOST 2 LAYERS decoded from position 0.  e5m2
#   codes are 12.5-25 % apart; where the two sides' fp32 sums (different summation orders) straddle a rounding boundary the cached
#   element differs by a whole code.  Measured on the 2-layer attention-true model (tools/fp8kv_study.py, profiles/r05_fp8kv.txt): 3.6e-4
#   .. 8.4e-4 of the cached elements differ, logits by median 1.2-2.0e-4, p99.9 2.7e-3, max 3.04e-3 over 4096 positions (largest at
#   short contexts, < 1.1e-3 beyond 256 positions: more rows to average over).  It is NOT depth-independent: every layer's cache
#   re-quantises a perturbed value (an fp32-rounding difference eps becomes ~0.18 sqrt(5.5 eps) behind one e5m2 row), and by 32 layers
#   ANY two summation orders sit at the fixed point, ~2e-2 -- the CPU checker against ITSELF with one norm weight moved by one ulp:
#   median 1.9e-2 (fp16 cache: 1.3e-4).  So full-depth fp8-KV comparisons run one step on identical caches (tests/test_long_context.py),
#   never free-running from position 0, and this constant is only ever applied to shallow models.
LOGIT_TOL = 1e-3
FP8KV_TOL = 4e-3


def logit_tol(kvbits: int) -> float:
    return FP8KV_TOL if kvbits == 8 else LOGIT_TOL


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from calm_amd.host import load_lib

        return load_lib().calm_hip_device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """a gpu-marked test on a box without a GPU is a hard error only when -m gpu was asked for;
    in a plain `pytest tests/` run on CPU it is skipped"""
    if _gpu_count() > 0:
        return
    asked = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if asked:
        return  # let them run and fail loudly: the HIP path must not silently disappear
    skip = pytest.mark.skip(reason="no HIP device on this box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hiplib():
    from calm_amd.host import load_lib

    return load_lib()


def rel_err(a: np.ndarray, ref: np.ndarray) -> float:
    """the parity metric: max |delta| over the row divided by max |ref| of the row
    (element-wise relative error is meaningless near zero logits: SURVEY.md appendix B.3)"""
    return float(np.abs(a.astype(np.float64) - ref.astype(np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def load_golden(name):
    from calm_amd.host import HostModel

    model = HostModel.from_file(os.path.join(GOLDEN, name + ".calm"))
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return model, z


def run_torchrun(script, world: int, env: dict, attempts: int = 3):
    """launch `script` under torch.distributed.run on 127.0.0.1 with a free port; a rendezvous that loses the race for
    its port (the probe socket is closed before torchrun binds it) is retried on a fresh port"""
    import socket
    import subprocess

    r = None
    for _ in range(attempts):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        r = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
            env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            break
    return r
