"""Host side of the decode path, mirroring the reference CLI's setup and loop (src/run.c).

    HostModel      = main()'s model set-up: metadata -> struct Config (run.c:32-69), tensor lookup
                     -> struct Weights (run.c:71-117), byte accounting (run.c:131-152,523-532)
    HipBackend     = the backend binding: upload_hip per "model.*" tensor (run.c:550-561),
                     prepare_hip (run.c:578-583), forward_hip through the C ABI (ctypes)
    generate()     = the greedy decode loop with the reference's throughput/bandwidth accounting
                     (run.c:167-256)

The arithmetic all happens behind the C ABI in libcalm_hip.so; this module is plumbing.  There is
deliberately NO CPU fallback here: without a GPU HipBackend raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import abi
from .calmfile import DBITS, CalmFile

_LIB = None
LIB_PATH = os.environ.get("CALM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcalm_hip.so")  # (CALM_HIP_LIB: tools/timeline.py's instrumented build)
TEST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcalm_hip_test.so")

# include/calm_hip.h: the drop-in library
EXPORTS = [
    "init_hip", "upload_hip", "alloc_hip", "prepare_hip", "forward_hip", "perf_hip", "calm_hip_device_count", "calm_hip_device_name", "calm_hip_configure", "calm_hip_query", "release_hip",
    "free_hip", "download_hip", "decode_greedy_hip", "decode_sample_hip", "prefill_hip", "prefill_logprobs_hip", "forward_stage_hip", "copy_hip", "perf_stage_hip",
]
# include/calm_hip_test.h: libcalm_hip_test.so, tests and tools only
TEST_EXPORTS = ["calm_hip_test_matvec", "calm_hip_test_norm_matvec", "calm_hip_test_attn", "calm_hip_test_argmax", "calm_hip_test_sample", "calm_hip_test_pf_gemm", "calm_hip_read_kv", "calm_hip_write_kv", "calm_hip_read_moe", "calm_hip_membench"]


class _Libs:
    """the drop-in library, plus -- looked up lazily, only when a test hook is asked for -- libcalm_hip_test.so"""

    def __init__(self, product: C.CDLL):
        self._product = product
        self._test = None

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return getattr(self._product, name)
        except AttributeError:
            if name not in TEST_EXPORTS:
                raise
        if self._test is None:
            if not os.path.exists(TEST_LIB_PATH):
                raise RuntimeError(f"{TEST_LIB_PATH} is missing: run `python -m calm_amd.build`")
            t = C.CDLL(TEST_LIB_PATH)
            T = C.POINTER(abi.Transformer)
            fp = C.POINTER(C.c_float)
            protos = {
                "calm_hip_test_matvec": (None, [C.c_int, C.c_void_p, fp, fp, C.c_int, C.c_int]),
                "calm_hip_test_norm_matvec": (None, [C.c_int, C.c_void_p, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_int]),
                "calm_hip_test_attn": (None, [fp, C.c_void_p, C.c_void_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
                "calm_hip_test_argmax": (C.c_int, [fp, C.c_int]),
                "calm_hip_test_sample": (C.c_int, [fp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_ulonglong)]),
                "calm_hip_test_pf_gemm": (None, [C.c_int, C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]),
                "calm_hip_read_kv": (None, [T, C.c_int, C.c_int, C.c_void_p]),
                "calm_hip_write_kv": (None, [T, C.c_int, C.c_int, C.c_void_p]),
                "calm_hip_read_moe": (None, [T, C.c_int, C.POINTER(C.c_int), fp]),
                "calm_hip_membench": (C.c_double, [C.c_size_t, C.c_int, C.c_int]),
            }
            for n, (res, args) in protos.items():
                fn = getattr(t, n)
                fn.restype = res
                fn.argtypes = args
            self._test = t
        return getattr(self._test, name)


def lib_loaded() -> bool:
    """has libcalm_hip.so (and with it the system HIP runtime) been brought up in this process?"""
    return _LIB is not None


def require_torch_first(what: str = "torch.distributed over RCCL") -> None:
    """A process that uses BOTH PyTorch-ROCm (for RCCL) and this library must initialise torch's device first: torch ships its own copy
    of the HIP runtime, and when /opt/rocm's copy (which libcalm_hip.so links) has claimed the device before it, torch finds none
    (profiles/r05_gpu_tests.txt section 2; the working order is exercised by tools/experiments/dist_probe.py, profiles/r06_dist_probe.txt).
    Fails with a message instead of a missing device."""
    if lib_loaded():
        raise RuntimeError(f"{what}: initialise torch (torch.cuda.set_device + init_process_group) BEFORE the first calm_amd.host.load_lib() / "
                           "HipBackend in this process -- libcalm_hip.so is already loaded (INTEGRATION.md section D)")


def load_lib() -> "_Libs":
    """dlopen libcalm_hip.so and declare the prototypes of include/calm_hip.h; the hooks of calm_hip_test.h resolve through the
    same object from libcalm_hip_test.so"""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m calm_amd.build` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(LIB_PATH)
    T = C.POINTER(abi.Transformer)
    fp = C.POINTER(C.c_float)
    protos = {
        "init_hip": (None, []),
        "upload_hip": (C.c_void_p, [C.c_void_p, C.c_size_t]),
        "alloc_hip": (C.c_void_p, [C.c_size_t]),
        "prepare_hip": (None, [T]),
        "forward_hip": (fp, [T, C.c_int, C.c_int, C.c_uint]),
        "perf_hip": (None, []),
        "calm_hip_device_count": (C.c_int, []),
        "calm_hip_device_name": (C.c_char_p, []),
        "calm_hip_configure": (C.c_int, [C.c_char_p, C.c_int]),
        "calm_hip_query": (C.c_int, [C.c_char_p, C.c_int]),
        "release_hip": (None, [T]),
        "free_hip": (None, [C.c_void_p]),
        "download_hip": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "decode_greedy_hip": (fp, [T, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
        "decode_sample_hip": (fp, [T, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(abi.Sampler)]),
        "prefill_hip": (None, [T, C.POINTER(C.c_int), C.c_int, C.c_int]),
        "prefill_logprobs_hip": (None, [T, C.POINTER(C.c_int), C.c_int, C.c_int, fp]),
        "forward_stage_hip": (fp, [T, C.c_int, C.c_int, C.c_uint, C.c_uint]),
        "copy_hip": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "perf_stage_hip": (C.c_double, [T, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # raises AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _LIB = _Libs(lib)
    return _LIB


STAGES = ["qkv", "attn", "attn_out", "ffn_up", "ffn_down", "output"]


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class HostModel:
    """a model held on the host: metadata + named tensors (numpy arrays or mmap views)"""

    def __init__(self, tensors: Dict[str, np.ndarray], metadata: Dict[str, object], context: int = 0):
        self.tensors = tensors
        self.metadata = {k: str(v) for k, v in metadata.items()}
        self.dtype = self.metadata["dtype"]
        self.dbits = DBITS[self.dtype]
        self.config = self._config(context)

    @classmethod
    def from_file(cls, path: str, context: int = 0) -> "HostModel":
        f = CalmFile(path)
        m = cls({n: f.tensor(n) for n in f.names()}, f.metadata, context)
        m._file = f
        return m

    # -- src/run.c:32-69
    def _config(self, context: int) -> abi.Config:
        md = self.metadata
        c = abi.Config()
        c.dim = int(md["dim"])
        c.hidden_dim = int(md["hidden_dim"])
        c.n_layers = int(md["n_layers"])
        c.n_heads = int(md["n_heads"])
        c.n_kv_heads = int(md["n_kv_heads"])
        c.vocab_size = int(md["vocab_size"])
        c.head_dim = int(md["head_dim"])
        msl = md.get("max_seq_len")
        c.seq_len = int(msl) if msl is not None and int(msl) < 4096 else 4096  # run.c:41-43: capped at 4096 unless -c
        if context:
            c.seq_len = context
        c.rope_theta = float(md["rope_theta"])
        c.rotary_dim = int(md["rotary_dim"])
        if "n_experts" in md:
            c.n_experts = int(md["n_experts"])
            c.n_experts_ac = int(md["n_experts_active"])
        c.norm_eps = float(md.get("norm_eps", 1e-5))
        c.act_gelu = md.get("act_type") == "gelu"
        nt = md.get("norm_type", "")
        c.norm_ln = nt.startswith("layernorm")
        c.norm_par = nt == "layernorm_par"
        c.qkv_clip = float(md["qkv_clip"]) if "qkv_clip" in md else float(np.finfo(np.float32).max)
        return c

    # -- src/run.c:131-152,523-532
    def accounting(self):
        def count(prefix, flt=None):
            b = p = 0
            for n, a in self.tensors.items():
                if not n.startswith(prefix) or (flt and flt not in n):
                    continue
                p += a.size * (8 if a.dtype == np.int32 else 1)
                b += a.nbytes
            return b, p

        n_bytes, n_params = count("model.")
        n_bw = n_bytes - count("model.embed.")[0]
        if "model.output.weight" not in self.tensors and "model.embed.weight" in self.tensors:
            n_bw += self.tensors["model.embed.weight"].nbytes
        if self.config.n_experts:
            mlp = count("model.layers.", ".mlp.w")[0]
            n_bw -= mlp
            n_bw += mlp // self.config.n_experts * self.config.n_experts_ac
        return n_params, n_bytes, n_bw

    def kv_bandwidth(self, kvbits: int, pos: int) -> int:
        """src/run.c:161-165"""
        c = self.config
        kv_len = c.seq_len if pos >= c.seq_len else pos + 1
        return 2 * (kvbits // 8) * c.n_layers * (c.head_dim * c.n_kv_heads) * kv_len

    # -- src/run.c:71-117 with `addr` supplying the pointer for each tensor (host or device)
    def fill_transformer(self, t: abi.Transformer, addr: Callable[[str], int], kvbits: int = 16) -> None:
        C.memmove(C.byref(t.config), C.byref(self.config), C.sizeof(abi.Config))
        w = t.weights
        w.dbits = self.dbits
        cfg = self.config
        has = lambda n: n in self.tensors
        if has("model.embed.weight"):  # absent on the later stages of a layer pipeline
            w.token_embedding_table = addr("model.embed.weight")
        for l in range(cfg.n_layers):
            p = f"model.layers.{l}."
            w.rms_att_weight[l] = addr(p + "attn.norm.weight")
            if not cfg.norm_par:
                w.rms_ffn_weight[l] = addr(p + "mlp.norm.weight")
            w.wq[l] = addr(p + "attn.wq.weight")
            w.wk[l] = addr(p + "attn.wk.weight")
            w.wv[l] = addr(p + "attn.wv.weight")
            w.wo[l] = addr(p + "attn.wo.weight")
            if has(p + "attn.wqkv.bias"):
                w.bqkv[l] = addr(p + "attn.wqkv.bias")
            if cfg.n_experts:
                w.moegate[l] = addr(p + "moegate.weight")
            w.w1[l] = addr(p + "mlp.w1.weight")
            w.w2[l] = addr(p + "mlp.w2.weight")
            w.w3[l] = addr(p + "mlp.w3.weight")
        if has("model.norm.weight"):  # absent on all but the last stage of a layer pipeline
            w.rms_final_weight = addr("model.norm.weight")
            w.wcls = addr("model.output.weight") if has("model.output.weight") else w.token_embedding_table
        t.state.kvbits = kvbits
        t.n_params, t.n_bytes, t.n_bandwidth = self.accounting()


class HipBackend:
    """binds a HostModel to libcalm_hip.so exactly the way src/run.c:550-596 binds a GPU backend"""

    def __init__(self, model: HostModel, kvbits: int = 16, stream=None, device_synth=None):
        """stream: optional iterable of (name, array) supplying the tensor bytes one at a time (buffers
        may be reused between items -- each is uploaded before the next is drawn); model.tensors then
        only needs shape/dtype placeholders (calmfile.stub_tensors).
        device_synth: (spec, dtype, seed, n_layers) -- the synthetic model of calmfile.synth_stream_big with those
        arguments, generated in device memory instead (calmfile.synth_device): nothing crosses PCIe but the small tensors"""
        self.lib = load_lib()
        if self.lib.calm_hip_device_count() <= 0:
            raise RuntimeError("no HIP device visible: the calm_amd backend has no CPU fallback")
        self.model = model
        self.kvbits = kvbits
        self.t = abi.Transformer()
        self._dev: Dict[str, int] = {}
        self._extra: List[int] = []
        self.lib.init_hip()
        # CALM_HIP_DEVICES=P > 1: the library splits the layers over P pipeline stages (include/calm_hip.h).  A host that knows
        # tensor names says where each tensor goes before it uploads or generates it; one that does not (run.c) lets upload_hip
        # defer to prepare_hip, for which the host arrays must stay alive until then.
        self.stages = self.lib.calm_hip_query(b"stages", 0)
        self._keep_host = []
        L, P = model.config.n_layers, self.stages

        def stage_of(name: str) -> int:
            if name.startswith("model.layers."):
                l, s0 = int(name.split(".")[2]), 0
                for st in range(P):
                    n = L // P + (1 if st < L % P else 0)
                    if l < s0 + n:
                        return st
                    s0 += n
            return 0 if name.startswith("model.embed.") else P - 1

        def route(name: str):
            if P > 1:
                st = stage_of(name) if name else 0
                self.lib.calm_hip_configure(b"stage", st)
                return st
            return None

        if device_synth is not None:
            from .calmfile import synth_device

            assert P == 1 or "model.output.weight" in model.tensors, "a tied classifier needs the embedding on two stages: upload it from the host"

            def upload(a):
                self._keep_host.append(a)
                return self.lib.upload_hip(a.ctypes.data, a.nbytes)

            gen = synth_device(*device_synth, alloc=self.lib.alloc_hip, upload=upload, before=route)
            for name, ptr in gen:
                if name:
                    self._dev[name] = ptr
                else:
                    self._extra.append(ptr)
        else:
            for name, a in (stream if stream is not None else model.tensors.items()):
                if name.startswith("model."):  # run.c:556-558
                    a = np.ascontiguousarray(a)
                    if P > 1:
                        if stream is not None:
                            route(name)  # streamed buffers are reused: place each tensor at once
                        else:
                            self._keep_host.append(a)  # deferred to prepare_hip, like run.c's mmap
                    self._dev[name] = self.lib.upload_hip(a.ctypes.data, a.nbytes)
        if P > 1:
            self.lib.calm_hip_configure(b"stage", -1)
        model.fill_transformer(self.t, lambda n: self._dev[n], kvbits)
        self.lib.prepare_hip(C.byref(self.t))
        self._keep_host = []
        self.vocab = model.config.vocab_size
        self._logits_addr, self._logits_view = 0, None

    def forward(self, token: int, pos: int, flags: int = 0) -> Optional[np.ndarray]:
        """-> view of the backend's logits buffer (valid until the next call), or None for KV-only"""
        p = self.lib.forward_hip(C.byref(self.t), token, pos, flags)
        if not p:
            return None
        # the backend's logits buffer never moves (pinned host memory, include/calm_hip.h): wrap it once
        addr = C.addressof(p.contents)
        if addr != self._logits_addr:
            self._logits_view = np.ctypeslib.as_array(p, shape=(self.vocab,))
            self._logits_addr = addr
        return self._logits_view

    def forward_stage(self, token: int, pos: int, flags: int, stage_flags: int) -> Optional[np.ndarray]:
        p = self.lib.forward_stage_hip(C.byref(self.t), token, pos, flags, stage_flags)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(self.vocab,))

    def export_x(self, dst_ptr: int) -> None:
        """state.x -> a caller buffer (device or host pointer)"""
        self.lib.copy_hip(dst_ptr, self.t.state.x, self.model.config.dim * 4)

    def import_x(self, src_ptr: int) -> None:
        self.lib.copy_hip(self.t.state.x, src_ptr, self.model.config.dim * 4)

    def decode_greedy(self, token: int, pos: int, n_steps: int):
        out = (C.c_int * n_steps)()
        p = self.lib.decode_greedy_hip(C.byref(self.t), token, pos, n_steps, out)
        return np.array(out[:], dtype=np.int64), np.ctypeslib.as_array(p, shape=(self.vocab,))

    def decode_sample(self, token: int, pos: int, n_steps: int, sampler: "abi.Sampler"):
        """n_steps tokens drawn on the device as src/sampler.c:80-90 draws them; `sampler.rng_state` is advanced"""
        out = (C.c_int * n_steps)()
        p = self.lib.decode_sample_hip(C.byref(self.t), token, pos, n_steps, out, C.byref(sampler))
        return np.array(out[:], dtype=np.int64), np.ctypeslib.as_array(p, shape=(self.vocab,))

    def prefill(self, tokens, pos: int) -> None:
        """KV-cache effect of forward(tokens[i], pos + i, FF_UPDATE_KV_ONLY) for every i, batched (prefill_hip)"""
        arr = (C.c_int * len(tokens))(*[int(t) for t in tokens])
        self.lib.prefill_hip(C.byref(self.t), arr, len(tokens), pos)

    def prefill_logprobs(self, tokens, pos: int) -> np.ndarray:
        """prefill + log softmax probability of every next token (len(tokens) - 1 values): perplexity's raw material"""
        arr = (C.c_int * len(tokens))(*[int(t) for t in tokens])
        out = np.zeros(max(len(tokens), 1), dtype=np.float32)
        self.lib.prefill_logprobs_hip(C.byref(self.t), arr, len(tokens), pos, fptr(out))
        return out[: max(len(tokens) - 1, 0)]

    def stage_us(self, stage: int, iters: int = 4):
        b = C.c_uint64(0)
        us = self.lib.perf_stage_hip(C.byref(self.t), stage, iters, C.byref(b))
        return us, b.value

    def read_state(self, field: str, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.float32)
        self.lib.download_hip(out.ctypes.data, getattr(self.t.state, field), out.nbytes)
        return out

    def read_kv(self, layer: int, which: int) -> np.ndarray:
        c = self.model.config
        out = np.empty((c.seq_len, c.head_dim * c.n_kv_heads), dtype=np.uint16)
        self.lib.calm_hip_read_kv(C.byref(self.t), layer, which, out.ctypes.data)
        return out.view(np.float16)

    def read_moe(self, layer: int):
        """routing of the last decode step at `layer`: (expert ids in rank order, their weights) -- test hook"""
        n = self.model.config.n_experts_ac
        e = (C.c_int * n)()
        w = np.empty(n, dtype=np.float32)
        self.lib.calm_hip_read_moe(C.byref(self.t), layer, e, fptr(w))
        return np.array(e[:], dtype=np.int64), w

    def write_kv(self, layer: int, which: int, rows: np.ndarray) -> None:
        """rows: (seq_len, kv_dim) float16 -> this layer's K (0) or V (1) cache (test hook)"""
        rows = np.ascontiguousarray(rows, dtype=np.float16)
        c = self.model.config
        assert rows.shape == (c.seq_len, c.head_dim * c.n_kv_heads)
        self.lib.calm_hip_write_kv(C.byref(self.t), layer, which, rows.ctypes.data)

    def close(self):
        if self.t is not None:
            self.lib.release_hip(C.byref(self.t))
            for d in list(self._dev.values()) + self._extra:
                self.lib.free_hip(d)
            self._dev.clear()
            self._extra = []
            self.t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def argmax_first(logits: np.ndarray) -> int:
    """greedy sampler of the reference (src/sampler.c:34-42): first index of the strict maximum"""
    return int(logits.argmax())


def generate(backend, model: HostModel, prompt_tokens: Sequence[int], steps: int, pos_offset: int = 0, kvbits: Optional[int] = None, batched_prompt: bool = False):
    """greedy decode loop of src/run.c:167-256 (temperature 0): returns (tokens, stats).

    The first len(prompt)-1 positions are KV-only prompt steps; timing covers the whole loop, and
    tok/s = positions / elapsed with prompt positions included, like the reference (run.c:249-253).
    batched_prompt: hand those positions to prefill_hip in one call instead;
    same tokens out.
    """
    FF = abi.FF_UPDATE_KV_ONLY
    kvbits = kvbits or getattr(backend, "kvbits", 16)  # the accounting follows the cache the backend really keeps (run.c:161-165)
    n_prompt = len(prompt_tokens)
    token = int(prompt_tokens[0])
    pos = 0
    out: List[int] = []
    forward = backend.forward  # (the loop body is on the clock: keep Python's share of a token small)
    t0 = time.perf_counter()
    if batched_prompt and n_prompt > 1 and steps >= n_prompt:
        backend.prefill(prompt_tokens[: n_prompt - 1], pos_offset)
        out.extend(int(t) for t in prompt_tokens[1:])
        pos = n_prompt - 1
        token = int(prompt_tokens[pos])
    while pos < steps:
        if pos < n_prompt - 1:
            forward(token, pos + pos_offset, FF)
            nxt = int(prompt_tokens[pos + 1])
        else:
            nxt = int(forward(token, pos + pos_offset, 0).argmax())  # src/sampler.c:34-42: first index of the maximum
        pos += 1
        out.append(nxt)
        token = nxt
    dt = time.perf_counter() - t0
    # the reference's bandwidth accounting (src/run.c:161-165,211-212), summed off the clock
    _, _, n_bw = model.accounting()
    read_bytes = sum(n_bw + model.kv_bandwidth(kvbits, p + pos_offset) for p in range(pos))
    stats = {"tokens": pos, "seconds": dt, "tok_s": pos / dt, "GBps": read_bytes / 1e9 / dt, "read_bytes": read_bytes}
    return out, stats


def perplexity(backend, tokens: Sequence[int], steps: int = 0):
    """the arithmetic of the reference's perplexity mode (study(), src/run.c:286-308) on top of
    backend.prefill_logprobs: positions wrap every `steps` tokens (0 = never), every window is scored in one
    batched call.  Returns (perplexity, standard error) exactly as the reference prints them."""
    import math

    n = len(tokens)
    logprobs: List[float] = []
    start = 0
    while start + 1 < n:
        end = n if steps <= 0 else min(n, start + steps)
        # the window's last token is scored against the first token of the next window (pos restarts at 0 there)
        window = list(tokens[start:end]) + ([tokens[end]] if end < n else [])
        lp = backend.prefill_logprobs(window, 0)
        logprobs.extend(float(v) for v in lp[: end - start if end < n else end - start - 1])
        start = end
    s = sum(logprobs)
    ss = sum(v * v for v in logprobs)
    den = float(len(logprobs))
    ppl = math.exp(-s / den)
    return ppl, ppl * math.sqrt(max(ss - s * s / den, 0.0) / den / den)
