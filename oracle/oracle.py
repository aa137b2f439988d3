"""ctypes loaders for the CPU checkers.  TEST INFRASTRUCTURE -- never imported by calm_amd/.

    OracleBackend : oracle/liboracle.so   -- our plain-C restatement of reference src/infer.c
    RefBackend    : oracle/_ref/libcalm_ref.so -- the reference's own src/infer.c, compiled untouched
                    (built by oracle/Makefile `ref` when /root/reference is present; travels to the
                    GPU box as a binary)

Both take a calm_amd.host.HostModel (host tensors) and expose forward(token, pos, flags) with the
reference's semantics (src/infer.c:311-472), plus per-op entry points for kernel unit tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from calm_amd import abi
from calm_amd.host import HostModel

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_ORACLE = os.path.join(HERE, "liboracle.so")
LIB_REF = os.path.join(HERE, "_ref", "libcalm_ref.so")
RUN_CPU = os.path.join(HERE, "_ref", "run_cpu")
RUN_HIP = os.path.join(HERE, "_ref", "run_hip")

_fp = C.POINTER(C.c_float)
_T = C.POINTER(abi.Transformer)


def ensure_built() -> None:
    if not os.path.exists(LIB_ORACLE) or os.path.getmtime(LIB_ORACLE) < os.path.getmtime(os.path.join(HERE, "calm_oracle.c")):
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, capture_output=True)


def have_ref() -> bool:
    return os.path.exists(LIB_REF)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        ensure_built()
        L = C.CDLL(LIB_ORACLE)
        L.oracle_half_to_float.restype = C.c_float
        L.oracle_half_to_float.argtypes = [C.c_uint16]
        L.oracle_float_to_half.restype = C.c_uint16
        L.oracle_float_to_half.argtypes = [C.c_float]
        L.oracle_float_to_e5m2.restype = C.c_uint8
        L.oracle_float_to_e5m2.argtypes = [C.c_float]
        L.oracle_decode_weight.restype = C.c_float
        L.oracle_decode_weight.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.oracle_matvec.restype = None
        L.oracle_matvec.argtypes = [_fp, _fp, C.c_void_p, _fp, C.c_int, C.c_int, C.c_int]
        L.oracle_norm.restype = None
        L.oracle_norm.argtypes = [_fp, _fp, _fp, C.c_int, C.c_float, C.c_int]
        L.oracle_rope.restype = None
        L.oracle_rope.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        L.oracle_attn_head.restype = None
        L.oracle_attn_head.argtypes = [_fp, _fp, _fp, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_act.restype = C.c_float
        L.oracle_act.argtypes = [C.c_float, C.c_int]
        L.oracle_moe_gate.restype = None
        L.oracle_moe_gate.argtypes = [_fp, C.POINTER(C.c_int), _fp, C.c_int, C.c_int]
        L.oracle_kv_slots.restype = None
        L.oracle_kv_slots.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_prepare.restype = None
        L.oracle_prepare.argtypes = [_T]
        L.oracle_release.restype = None
        L.oracle_release.argtypes = [_T]
        L.oracle_forward.restype = _fp
        L.oracle_forward.argtypes = [_T, C.c_int, C.c_int, C.c_uint]
        L.oracle_forward_stage.restype = _fp
        L.oracle_forward_stage.argtypes = [_T, C.c_int, C.c_int, C.c_uint, C.c_uint]
        L.oracle_trace_moe.restype = None
        L.oracle_trace_moe.argtypes = [C.POINTER(C.c_int), _fp, _fp]
        L.oracle_argmax.restype = C.c_int
        L.oracle_argmax.argtypes = [_fp, C.c_int]
        L.oracle_sample.restype = C.c_int
        L.oracle_sample.argtypes = [_fp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_ulonglong)]
        _lib = L
    return _lib


def _f(a: np.ndarray):
    return a.ctypes.data_as(_fp)


# ---- per-op helpers (numpy in / numpy out) ------------------------------------------------------

def matvec(w: np.ndarray, x: np.ndarray, dbits: int, n: int, d: int, bias: Optional[np.ndarray] = None) -> np.ndarray:
    out = np.empty(d, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w)
    lib().oracle_matvec(_f(out), _f(x), w.ctypes.data, _f(bias) if bias is not None else None, n, d, dbits)
    return out


def norm(x: np.ndarray, weight: np.ndarray, eps: float, ln: bool) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_norm(_f(out), _f(x), _f(weight), x.size, eps, int(ln))
    return out


def attention(q: np.ndarray, kcache: np.ndarray, vcache: np.ndarray, n_heads: int, n_kv_heads: int, head_dim: int, kv_len: int) -> np.ndarray:
    """q (n_heads*head_dim) fp32; caches fp16 [seq_len][kv_dim]; src/infer.c:397-406"""
    kv_dim = n_kv_heads * head_dim
    kv_mul = n_heads // n_kv_heads
    out = np.empty(n_heads * head_dim, dtype=np.float32)
    att = np.empty(kv_len + 1, dtype=np.float32)
    k16 = np.ascontiguousarray(kcache).view(np.uint16)
    v16 = np.ascontiguousarray(vcache).view(np.uint16)
    q = np.ascontiguousarray(q, dtype=np.float32)
    for h in range(n_heads):
        off = (h // kv_mul) * head_dim * 2
        lib().oracle_attn_head(
            C.cast(out.ctypes.data + h * head_dim * 4, _fp), _f(att), C.cast(q.ctypes.data + h * head_dim * 4, _fp),
            k16.ctypes.data + off, v16.ctypes.data + off, head_dim, kv_dim, kv_len)
    return out


def moe_gate(logits: np.ndarray, active: int):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    w = np.empty(active, dtype=np.float32)
    e = (C.c_int * active)()
    lib().oracle_moe_gate(_f(w), e, _f(logits), logits.size, active)
    return w, np.array(e[:])


def argmax(logits: np.ndarray) -> int:
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    return lib().oracle_argmax(_f(logits), logits.size)


def sample(logits: np.ndarray, temperature: float, minp: float, rng_state: int):
    """one draw of the host sampler (src/sampler.c:80-90) -> (token, advanced rng state)"""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    st = C.c_ulonglong(rng_state)
    tok = lib().oracle_sample(_f(logits), logits.size, temperature, minp, C.byref(st))
    return tok, st.value


# ---- whole-model backends ----------------------------------------------------------------------

class _CpuBackend:
    def __init__(self, model: HostModel, prepare, forward, release=None, kvbits: int = 16):
        self.model = model
        self.kvbits = kvbits
        self.t = abi.Transformer()
        self._keep = {n: np.ascontiguousarray(a) for n, a in model.tensors.items() if n.startswith("model.")}
        model.fill_transformer(self.t, lambda n: self._keep[n].ctypes.data, kvbits)
        self._forward = forward
        self._release = release
        prepare(C.byref(self.t))
        self.vocab = model.config.vocab_size

    def forward(self, token: int, pos: int, flags: int = 0):
        p = self._forward(C.byref(self.t), token, pos, flags)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(self.vocab,))

    def prefill(self, tokens, pos: int) -> None:
        """the reference's serial prompt loop (src/run.c:204-218): what prefill_hip must be equivalent to"""
        for i, tok in enumerate(tokens):
            self._forward(C.byref(self.t), int(tok), pos + i, abi.FF_UPDATE_KV_ONLY)

    def prefill_logprobs(self, tokens, pos: int) -> np.ndarray:
        """the reference's perplexity inner loop (src/run.c:294-298; sample_prob, src/sampler.c:19-32), one forward per token"""
        out = np.zeros(max(len(tokens) - 1, 0), dtype=np.float32)
        for i, tok in enumerate(tokens):
            lg = self.forward(int(tok), pos + i, 0)
            if i + 1 < len(tokens):
                e = np.exp(lg - lg.max(), dtype=np.float32)
                out[i] = np.log(np.float64(e[int(tokens[i + 1])] / e.sum(dtype=np.float32)))
        return out

    def kv(self, layer: int, which: int) -> np.ndarray:
        c = self.model.config
        kv_dim = c.head_dim * c.n_kv_heads
        base = self.t.state.value_cache if which else self.t.state.key_cache
        n = c.seq_len * kv_dim
        arr = np.ctypeslib.as_array(C.cast(base + layer * n * 2, C.POINTER(C.c_uint16)), shape=(c.seq_len, kv_dim))
        return arr.view(np.float16)

    def state(self, field: str, count: int) -> np.ndarray:
        return np.ctypeslib.as_array(C.cast(getattr(self.t.state, field), _fp), shape=(count,))

    def close(self):
        if self._release and self.t is not None:
            self._release(C.byref(self.t))
        self.t = None


class OracleBackend(_CpuBackend):
    """our C restatement (liboracle.so)"""

    def __init__(self, model: HostModel, kvbits: int = 16):
        """kvbits = 8: K/V rows (and the re-rotated sink keys) are rounded to fp8 e5m2 as the reference's CUDA backend stores
        them (src/infer.cu:473-482,150-180; calm_oracle.c: kv_store) -- the checker of the HIP backend's fp8 cache"""
        L = lib()
        super().__init__(model, L.oracle_prepare, L.oracle_forward, L.oracle_release, kvbits)

    def forward_traced(self, token: int, pos: int):
        """forward() of a mixture-of-experts model that also returns every layer's routing:
        (logits, experts [n_layers][n_active], weights [n_layers][n_active], gate logits [n_layers][n_experts])"""
        c = self.model.config
        e = np.zeros((c.n_layers, c.n_experts_ac), dtype=np.int32)
        w = np.zeros((c.n_layers, c.n_experts_ac), dtype=np.float32)
        g = np.zeros((c.n_layers, c.n_experts), dtype=np.float32)
        L = lib()
        L.oracle_trace_moe(e.ctypes.data_as(C.POINTER(C.c_int)), _f(w), _f(g))
        try:
            lg = self.forward(token, pos, 0)
        finally:
            L.oracle_trace_moe(None, None, None)
        return lg, e, w, g

    # the pipeline-stage surface of calm_amd.host.HipBackend, on host memory (tests of the stage protocol)
    def forward_stage(self, token: int, pos: int, flags: int, stage_flags: int):
        p = lib().oracle_forward_stage(C.byref(self.t), token, pos, flags, stage_flags)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(self.vocab,))

    def export_x(self, dst_ptr: int) -> None:
        C.memmove(dst_ptr, self.t.state.x, self.model.config.dim * 4)

    def import_x(self, src_ptr: int) -> None:
        C.memmove(self.t.state.x, src_ptr, self.model.config.dim * 4)


_ref = None


def ref_lib() -> C.CDLL:
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libcalm_ref.so not built (needs /root/reference): make -C oracle ref")
        R = C.CDLL(LIB_REF)
        R.prepare.restype = None
        R.prepare.argtypes = [_T]
        R.forward.restype = _fp
        R.forward.argtypes = [_T, C.c_int, C.c_int, C.c_uint]
        _ref = R
    return _ref


class RefBackend(_CpuBackend):
    """the reference's own CPU backend: src/infer.c prepare()/forward(), compiled untouched"""

    def __init__(self, model: HostModel):
        R = ref_lib()
        super().__init__(model, R.prepare, R.forward, None)
