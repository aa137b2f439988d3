"""`.calm` model files: reader, writer, quantisers and synthetic-model generator.

A .calm file is a safetensors container with calm's metadata keys and tensor names; the on-disk
contract is defined by its consumers -- reference src/tensors.c:216-270 (parser),
src/run.c:32-117 (metadata keys, tensor names/shapes) -- and its producer tools/convert.py
(:55-125 metadata, :245-268 gf4, :502-536 writer).  Nothing here is on the decode hot path: this
module exists because there are no real checkpoints on the build or GPU boxes, so tests and
bench.py synthesise models of the real SHAPES with seeded random weights.
"""
from __future__ import annotations

import dataclasses
import json
import math
import mmap
import struct
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

# dtype tags understood by the reference parser (src/tensors.c:69-95) that we emit / read
_NP2TAG = {np.dtype(np.float32): "F32", np.dtype(np.float16): "F16", np.dtype(np.int32): "I32", np.dtype(np.uint8): "U8"}
_TAG2NP = {"F32": np.float32, "F16": np.float16, "I32": np.int32, "U8": np.uint8, "F8_E5M2": np.uint8, "I8": np.int8, "I16": np.int16}
_ALIGN = 256  # data area starts 256-byte aligned (tools/convert.py:514,526)


class Fp8Bytes(np.ndarray):
    """uint8 array whose bytes are fp8 e5m2 codes (written with dtype tag F8_E5M2)."""


def as_fp8(a: np.ndarray) -> "Fp8Bytes":
    assert a.dtype == np.uint8
    return a.view(Fp8Bytes)


# ------------------------------------------------------------------------------------------------
# quantisers
# ------------------------------------------------------------------------------------------------

def fp8_e5m2_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint16) << 8).view(np.float16).astype(np.float32)


def quantize_gf4(w: np.ndarray) -> np.ndarray:
    """float32 (..., n) with n % 8 == 0 -> int32 (..., n/8) gf4 words.

    Format (tools/convert.py:245-268, decoded by src/infer.c:37-40): per group of 8 values the
    signed max-magnitude element, rounded to fp8 e5m2, is the scale S (bits 0-7); value k is
    stored as a 3-bit code q_k = clamp(round(v_k / S * -4 + 4), 0, 7) at bits 8+3k, and decodes
    to (q_k - 4) * S / -4.  The max element itself always encodes as q = 0 (-> +S).
    """
    assert w.shape[-1] % 8 == 0
    g = w.astype(np.float32).reshape(w.shape[:-1] + (w.shape[-1] // 8, 8))
    idx = np.abs(g).argmax(-1)
    gmax = np.take_along_axis(g, idx[..., None], -1)
    s_code = f32_to_fp8_e5m2(gmax)
    s = fp8_e5m2_to_f32(s_code)
    with np.errstate(divide="ignore", invalid="ignore"):
        nrm = g / s
    nrm = np.nan_to_num(nrm, nan=0.0, posinf=0.0, neginf=0.0)
    # the producer evaluates (x.half() * -4 + 4) in fp16, clamps, rounds half-to-even
    q = nrm.astype(np.float16) * np.float16(-4) + np.float16(4)
    q = np.rint(np.clip(q, 0, 7).astype(np.float32)).astype(np.int64)
    shifts = np.array([8 + 3 * k for k in range(8)], dtype=np.int64)
    word = (q << shifts).sum(-1) + s_code[..., 0].astype(np.int64)
    return (word & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def f32_to_fp8_e5m2(a: np.ndarray) -> np.ndarray:
    """float32 -> fp8 e5m2 codes (uint8): single-step round-to-nearest-even, overflow to inf -- what
    torch's .to(float8_e5m2) does in tools/convert.py:307-311 -- in integer arithmetic.  e5m2 is the
    top byte of the binary16 pattern (reference src/infer.c:28-35)."""
    x = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    sign = ((x >> 24) & 0x80).astype(np.uint64)
    absx = x & 0x7FFFFFFF
    out = np.zeros(absx.shape, dtype=np.uint64)
    e = (absx >> 23).astype(np.int64) - 127
    man = (absx & 0x7FFFFF) | 0x800000
    # normal e5m2 range: e in [-14, 15]; subnormal: e in [-17(ish), -15]
    shift = np.where(e < -14, 21 + (-14 - e), 21)
    shift = np.clip(shift, 0, 40)
    base = np.where(e < -14, 0, (e + 15) << 2).astype(np.int64)
    manq = np.where(e < -14, man, man & 0x7FFFFF).astype(np.uint64)
    q = manq >> shift.astype(np.uint64)
    rem = manq & ((np.uint64(1) << shift.astype(np.uint64)) - np.uint64(1))
    half = np.uint64(1) << (shift.astype(np.uint64) - np.uint64(1))
    q = q + ((rem > half) | ((rem == half) & ((q & np.uint64(1)) == 1))).astype(np.uint64)
    val = base.astype(np.uint64) + q
    val = np.minimum(val, 0x7C)  # overflow -> inf
    out = np.where(absx == 0, 0, val)
    is_inf = absx == 0x7F800000
    is_nan = absx > 0x7F800000
    out = np.where(is_inf, 0x7C, out)
    out = np.where(is_nan, 0x7F, out)
    return (out | sign).astype(np.uint8)


def gf4_to_f32(words: np.ndarray) -> np.ndarray:
    """int32 (..., m) gf4 words -> float32 (..., 8m)   (reference src/infer.c:37-40)"""
    v = words.view(np.uint32).astype(np.uint64)
    s = fp8_e5m2_to_f32((v & 0xFF).astype(np.uint8)) / np.float32(-4)
    ks = np.arange(8, dtype=np.uint64)
    q = ((v[..., None] >> (8 + 3 * ks)) & 7).astype(np.int32) - 4
    return (q.astype(np.float32) * s[..., None]).reshape(words.shape[:-1] + (words.shape[-1] * 8,))


def quantize(w: np.ndarray, dtype: str) -> np.ndarray:
    """float32 weights -> storage array for dtype in {fp16, fp8, gf4}"""
    if dtype == "fp16":
        return w.astype(np.float16)
    if dtype == "fp8":
        return as_fp8(f32_to_fp8_e5m2(w))
    if dtype == "gf4":
        return quantize_gf4(w)
    raise ValueError(dtype)


def dequantize(a: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "fp16":
        return a.view(np.float16).astype(np.float32)
    if dtype == "fp8":
        return fp8_e5m2_to_f32(a.view(np.uint8))
    if dtype == "gf4":
        return gf4_to_f32(a)
    raise ValueError(dtype)


DBITS = {"fp16": 16, "fp8": 8, "gf4": 4}

# ------------------------------------------------------------------------------------------------
# container I/O
# ------------------------------------------------------------------------------------------------

def _tag(a: np.ndarray) -> str:
    if isinstance(a, Fp8Bytes):
        return "F8_E5M2"
    return _NP2TAG[a.dtype]


def write_calm(path: str, tensors: Dict[str, np.ndarray], metadata: Dict[str, object]) -> None:
    """safetensors layout: u64 LE header size, JSON header space-padded so the data area is
    256-byte aligned, then raw tensor bytes in header order (tools/convert.py:502-536).
    Metadata values are written as strings (src/tensors.c:184-214 accepts string->string only)."""
    header = {"__metadata__": {k: str(v) for k, v in metadata.items()}}
    off = 0
    for name, a in tensors.items():
        a = np.ascontiguousarray(a)
        header[name] = {"dtype": _tag(tensors[name]), "shape": list(a.shape), "data_offsets": [off, off + a.nbytes]}
        off += a.nbytes
    hjson = json.dumps(header).encode("utf-8")
    assert b"\\" not in hjson, "the reference parser rejects backslashes (src/tensors.c:31)"
    hjson += b" " * (-(len(hjson) + 8) % _ALIGN)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hjson)))
        f.write(hjson)
        for name, a in tensors.items():
            np.ascontiguousarray(a).view(np.uint8).reshape(-1).tofile(f)


def write_calm_stream(path: str, layout: Dict[str, np.ndarray], stream: Iterable[Tuple[str, np.ndarray]], metadata: Dict[str, object]) -> int:
    """write_calm for models that should not sit in host memory: the header is laid out from `layout` (name ->
    anything with the tensor's shape / dtype / nbytes, e.g. stub_tensors' placeholders), the bytes then come one
    tensor at a time from `stream`, in the same order, and go straight to the file (the reference converter holds the
    whole model, tools/convert.py:502-536).  Returns the file size."""
    header = {"__metadata__": {k: str(v) for k, v in metadata.items()}}
    off = 0
    for name, a in layout.items():
        header[name] = {"dtype": _tag(a), "shape": list(a.shape), "data_offsets": [off, off + a.nbytes]}
        off += a.nbytes
    hjson = json.dumps(header).encode("utf-8")
    assert b"\\" not in hjson, "the reference parser rejects backslashes (src/tensors.c:31)"
    hjson += b" " * (-(len(hjson) + 8) % _ALIGN)
    names = iter(layout.items())
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hjson)))
        f.write(hjson)
        for name, a in stream:
            want_name, want = next(names)
            if name != want_name or tuple(a.shape) != tuple(want.shape) or _tag(a) != _tag(want):
                raise ValueError(f"stream item {name} {a.shape} {_tag(a)} does not match the layout's {want_name} {want.shape} {_tag(want)}")
            np.ascontiguousarray(a).view(np.uint8).reshape(-1).tofile(f)
        if next(names, None) is not None:
            raise ValueError("stream ended before the layout did")
        return f.tell()


class CalmFile:
    """read-only view of a .calm file (mmap); tensors come back as numpy views"""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        (hsize,) = struct.unpack("<Q", self._mm[:8])
        header = json.loads(self._mm[8 : 8 + hsize].decode("utf-8"))
        self.metadata: Dict[str, str] = header.pop("__metadata__", {})
        self._base = 8 + hsize
        self.entries: Dict[str, Tuple[str, Tuple[int, ...], int, int]] = {}
        for name, e in header.items():
            b, en = e["data_offsets"]
            self.entries[name] = (e["dtype"], tuple(e["shape"]), b, en)

    def names(self) -> Iterable[str]:
        return self.entries.keys()

    def has(self, name: str) -> bool:
        return name in self.entries

    def dtype_tag(self, name: str) -> str:
        return self.entries[name][0]

    def nbytes(self, name: str) -> int:
        _, _, b, e = self.entries[name]
        return e - b

    def tensor(self, name: str) -> np.ndarray:
        tag, shape, b, e = self.entries[name]
        a = np.frombuffer(self._mm, dtype=_TAG2NP[tag], count=int(np.prod(shape)) if shape else 1, offset=self._base + b)
        return a.reshape(shape)

    def close(self):
        self._mm.close()
        self._f.close()


# ------------------------------------------------------------------------------------------------
# model shapes
# ------------------------------------------------------------------------------------------------

@dataclasses.dataclass
class ModelSpec:
    """architecture hyper-parameters = the .calm metadata keys (src/run.c:32-69)"""

    name: str
    dim: int
    hidden_dim: int
    head_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    rope_theta: float = 10000.0
    rotary_dim: Optional[int] = None
    max_seq_len: int = 4096
    n_experts: int = 0
    n_experts_active: int = 0
    norm_eps: float = 1e-5
    act_type: str = "silu"
    norm_type: str = "rmsnorm"
    qkv_clip: Optional[float] = None
    qkv_bias: bool = False
    tied: bool = False

    def __post_init__(self):
        if self.rotary_dim is None:
            self.rotary_dim = self.head_dim

    @property
    def q_dim(self):
        return self.n_heads * self.head_dim

    @property
    def kv_dim(self):
        return self.n_kv_heads * self.head_dim

    def metadata(self, dtype: str) -> Dict[str, object]:
        md: Dict[str, object] = {
            "arch": self.name,
            "dtype": dtype,
            "dim": self.dim,
            "hidden_dim": self.hidden_dim,
            "head_dim": self.head_dim,
            "n_layers": self.n_layers,
            "n_heads": self.n_heads,
            "n_kv_heads": self.n_kv_heads,
            "vocab_size": self.vocab_size,
            "max_seq_len": self.max_seq_len,
            "rope_theta": self.rope_theta,
            "rotary_dim": self.rotary_dim,
            "norm_eps": self.norm_eps,
            "norm_type": self.norm_type,
            "act_type": self.act_type,
            # -1: no BOS is prepended and decode never stops early (src/run.c:225, tokenizer.c:207)
            "bos_token_id": -1,
            "eos_token_id": -1,
        }
        if self.n_experts:
            md["n_experts"] = self.n_experts
            md["n_experts_active"] = self.n_experts_active
        if self.qkv_clip is not None:
            md["qkv_clip"] = self.qkv_clip
        return md


# the BASELINE.json configs (public HF shapes)
SPECS = {
    "tinyllama-1.1b": ModelSpec("tinyllama-1.1b", 2048, 5632, 64, 22, 32, 4, 32000, 1e4, max_seq_len=2048),
    "mistral-7b": ModelSpec("mistral-7b", 4096, 14336, 128, 32, 32, 8, 32000, 1e6, max_seq_len=32768),
    "llama-3-8b": ModelSpec("llama-3-8b", 4096, 14336, 128, 32, 32, 8, 128256, 5e5, max_seq_len=8192),
    "mixtral-8x7b": ModelSpec("mixtral-8x7b", 4096, 14336, 128, 32, 32, 8, 32000, 1e6, max_seq_len=32768, n_experts=8, n_experts_active=2),
    "dbrx-132b": ModelSpec("dbrx-132b", 6144, 10752, 128, 40, 48, 8, 100352, 5e5, max_seq_len=32768, n_experts=16, n_experts_active=4,
                           norm_type="layernorm", qkv_clip=8.0),
}


# Shapes of further public architectures the reference's converter emits (tools/convert.py:58-125: llama, qwen2, phi3, cohere, olmoe,
# gemma ...; config.json values of the public checkpoints) -- tests/test_shape_sweep.py runs 1-2 layers of each at full width: the
# launchers' grid / tile-shape rules were tuned on the five BASELINE shapes above, these are the sixth to thirteenth.
ARCH_SPECS = {
    "llama-2-7b": ModelSpec("llama-2-7b", 4096, 11008, 128, 32, 32, 32, 32000, 1e4, max_seq_len=4096),                      # MHA: kv_mul 1
    "llama-2-13b": ModelSpec("llama-2-13b", 5120, 13824, 128, 40, 40, 40, 32000, 1e4, max_seq_len=4096),                    # 5-KiB fp8 rows
    "yi-34b": ModelSpec("yi-34b", 7168, 20480, 128, 60, 56, 8, 64000, 5e6, max_seq_len=4096),                               # kv_mul 7
    "qwen2-7b": ModelSpec("qwen2-7b", 3584, 18944, 128, 28, 28, 4, 152064, 1e6, max_seq_len=32768, norm_eps=1e-6, qkv_bias=True),  # ragged 3.5-KiB rows, bias
    "phi-3-mini": ModelSpec("phi-3-mini", 3072, 8192, 96, 32, 32, 32, 32064, 1e4, max_seq_len=2048),                        # head size 96
    "command-r-35b": ModelSpec("command-r-35b", 8192, 22528, 128, 40, 64, 64, 256000, 8e6, max_seq_len=8192, norm_type="layernorm_par", tied=True),
    "gemma-7b": ModelSpec("gemma-7b", 3072, 24576, 256, 28, 16, 16, 256000, 1e4, max_seq_len=8192, norm_eps=1e-6, act_type="gelu", tied=True),  # head size 256, q_dim > dim
    "olmoe-1b-7b": ModelSpec("olmoe-1b-7b", 2048, 1024, 128, 16, 16, 16, 50304, 1e4, max_seq_len=4096, n_experts=64, n_experts_active=8),
}


def tiny_spec(name="tiny", **kw) -> ModelSpec:
    """small shapes for unit parity (seconds on the CPU oracle)"""
    base = dict(dim=64, hidden_dim=160, head_dim=16, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=320, rope_theta=1e4, max_seq_len=64)
    base.update(kw)
    return ModelSpec(name, **base)


# ------------------------------------------------------------------------------------------------
# synthetic weights
# ------------------------------------------------------------------------------------------------

def _toy_tokenizer(vocab: int) -> Tuple[np.ndarray, np.ndarray]:
    """token 0 '<unk>', printable ASCII as themselves where they fit, the rest '[id]';
    all strings unique, NUL-terminated, concatenated (tools/convert.py:492-495)."""
    toks = []
    for i in range(vocab):
        if i == 0:
            toks.append("<unk>")
        elif 32 <= i < 127 and chr(i) not in "\\\"":
            toks.append(chr(i))
        else:
            toks.append(f"[{i}]")
    blob = b"".join(t.encode("utf-8") + b"\0" for t in toks)
    return np.frombuffer(blob, dtype=np.uint8).copy(), np.zeros(vocab, dtype=np.float32)


def _rand_weight(rng: np.random.Generator, shape, sigma: float) -> np.ndarray:
    """N(0, sigma^2) rounded to fp16 (the converter casts the checkpoint through fp16 first,
    tools/convert.py:307-311), returned as float32"""
    w = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(sigma)
    return w.astype(np.float16).astype(np.float32)


def synth_tensors(spec: ModelSpec, dtype: str, seed: int = 0, sigma: Optional[float] = None, embed_sigma: float = 1.0,
                  norm_jitter: float = 0.1, layers: Optional[Iterable[int]] = None):
    """yield (name, array) for every tensor of a synthetic model, in file order.

    Weight scale: sigma defaults to 1/sqrt(fan_in) so activations stay O(1) through any depth
    (keeps logits well away from fp32 overflow and the softmax/top-k decisions well conditioned).
    Norm weights are 1 +- norm_jitter so a swapped or skipped norm weight is visible in parity.
    """
    rng = np.random.default_rng(seed)
    s = spec
    E = s.n_experts

    def W(rows, cols, fan_in=None, scale=1.0, lead=()):
        sg = (sigma if sigma is not None else 1.0 / math.sqrt(fan_in or cols)) * scale
        return quantize(_rand_weight(rng, lead + (rows, cols), sg), dtype)

    yield "model.embed.weight", quantize(_rand_weight(rng, (s.vocab_size, s.dim), embed_sigma), dtype)
    for l in (range(s.n_layers) if layers is None else layers):
        p = f"model.layers.{l}."
        yield p + "attn.norm.weight", (1 + norm_jitter * rng.standard_normal(s.dim)).astype(np.float32)
        yield p + "attn.wq.weight", W(s.q_dim, s.dim)
        yield p + "attn.wk.weight", W(s.kv_dim, s.dim)
        yield p + "attn.wv.weight", W(s.kv_dim, s.dim)
        yield p + "attn.wo.weight", W(s.dim, s.q_dim)
        if s.qkv_bias:
            yield p + "attn.wqkv.bias", (0.1 * rng.standard_normal(s.q_dim + 2 * s.kv_dim)).astype(np.float32)
        if s.norm_type != "layernorm_par":
            yield p + "mlp.norm.weight", (1 + norm_jitter * rng.standard_normal(s.dim)).astype(np.float32)
        if E:
            yield p + "moegate.weight", W(E, s.dim, scale=4.0)  # spread the gate logits: top-k far from ties
            yield p + "mlp.w1.weight", W(s.hidden_dim, s.dim, lead=(E,))
            yield p + "mlp.w2.weight", W(s.dim, s.hidden_dim, lead=(E,))
            yield p + "mlp.w3.weight", W(s.hidden_dim, s.dim, lead=(E,))
        else:
            yield p + "mlp.w1.weight", W(s.hidden_dim, s.dim)
            yield p + "mlp.w2.weight", W(s.dim, s.hidden_dim)
            yield p + "mlp.w3.weight", W(s.hidden_dim, s.dim)
    yield "model.norm.weight", (1 + norm_jitter * rng.standard_normal(s.dim)).astype(np.float32)
    if not s.tied:
        yield "model.output.weight", W(s.vocab_size, s.dim)
    toks, scores = _toy_tokenizer(s.vocab_size)
    yield "tokenizer.tokens", toks
    yield "tokenizer.scores", scores


def synth_model(spec: ModelSpec, dtype: str, seed: int = 0, **kw) -> Tuple[Dict[str, np.ndarray], Dict[str, object]]:
    return dict(synth_tensors(spec, dtype, seed, **kw)), spec.metadata(dtype)


def write_synth(path: str, spec: ModelSpec, dtype: str, seed: int = 0, **kw) -> None:
    tensors, md = synth_model(spec, dtype, seed, **kw)
    write_calm(path, tensors, md)


def spec_accounting(spec: ModelSpec, dtype: str) -> Dict[str, int]:
    """n_params / n_bytes / n_bandwidth of src/run.c:131-152,523-532 from the shapes alone (no weights)"""
    bits = DBITS[dtype]
    s = spec
    E = max(s.n_experts, 1)

    def wb(rows, cols):
        return rows * cols * bits // 8

    embed = wb(s.vocab_size, s.dim)
    per_layer_attn = wb(s.q_dim, s.dim) + 2 * wb(s.kv_dim, s.dim) + wb(s.dim, s.q_dim)
    per_layer_mlp = 3 * wb(s.hidden_dim, s.dim) * E
    per_layer_f32 = 4 * s.dim * (1 if s.norm_type == "layernorm_par" else 2) + (4 * (s.q_dim + 2 * s.kv_dim) if s.qkv_bias else 0)
    gate = wb(s.n_experts, s.dim) if s.n_experts else 0
    out = 0 if s.tied else wb(s.vocab_size, s.dim)
    n_bytes = embed + s.n_layers * (per_layer_attn + per_layer_mlp + per_layer_f32 + gate) + 4 * s.dim + out
    n_bw = n_bytes - embed + (embed if s.tied else 0)
    if s.n_experts:
        mlp = s.n_layers * per_layer_mlp
        n_bw = n_bw - mlp + mlp // s.n_experts * s.n_experts_active
    wparams = (s.vocab_size * s.dim * (1 if s.tied else 2)
               + s.n_layers * (s.q_dim * s.dim * 2 + 2 * s.kv_dim * s.dim + 3 * s.hidden_dim * s.dim * E + (s.n_experts * s.dim if s.n_experts else 0)))
    fparams = s.n_layers * per_layer_f32 // 4 + s.dim
    return {"n_params": wparams + fparams, "n_bytes": n_bytes, "n_bandwidth": n_bw}


# ------------------------------------------------------------------------------------------------
# full-size synthetic models (bench / full-size parity): seconds, not minutes
# ------------------------------------------------------------------------------------------------

_ICDF: Optional[np.ndarray] = None


_ICDF_T4 = None


def _icdf_table(tail: str = "normal") -> np.ndarray:
    """unit-variance quantiles at the 2^16 mid-points of (0, 1): "normal", or "t4" = Student-t with 4 degrees of freedom (variance 2)
    divided by sqrt(2) -- the heavy-tailed weights of synth_model_big(outliers=True): 1 in 10^4 entries beyond 5 sigma (normal: 1 in 10^6)"""
    global _ICDF_T4
    if tail == "normal":
        return _normal_icdf_table()
    assert tail == "t4", tail
    if _ICDF_T4 is None:
        from scipy.stats import t as student

        u = (np.arange(1 << 16) + 0.5) / (1 << 16)
        _ICDF_T4 = student.ppf(u, 4) / math.sqrt(2.0)
    return _ICDF_T4


def _normal_icdf_table(bits: int = 16) -> np.ndarray:
    """standard-normal quantiles at the 2^bits mid-points of (0, 1): u -> z, as float64"""
    global _ICDF
    if _ICDF is None:
        n = 1 << bits
        # invert the CDF by interpolation on a fine exact grid (monotone; the error is far below the
        # fp16 rounding applied afterwards)
        z = np.linspace(-4.6, 4.6, 200001)
        cdf = 0.5 * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0)))
        u = (np.arange(n) + 0.5) / n
        _ICDF = np.interp(u, cdf, z)
    return _ICDF


_LUT_CACHE: Dict[Tuple, np.ndarray] = {}


def _code_lut(dtype: str, sigma: float, tail: str = "normal") -> np.ndarray:
    """65536-entry table: uniform u16 -> storage code of a N(0, sigma^2) sample (fp16 bits or e5m2 byte); tail: _icdf_table"""
    key = (dtype, float(sigma), tail)
    if key not in _LUT_CACHE:
        z = (_icdf_table(tail) * sigma).astype(np.float32).astype(np.float16)
        if dtype == "fp16":
            _LUT_CACHE[key] = z.view(np.uint16).copy()
        else:
            _LUT_CACHE[key] = f32_to_fp8_e5m2(z.astype(np.float32))
    return _LUT_CACHE[key]


def _gf4_scale_lut(sigma: float, tail: str = "normal") -> np.ndarray:
    """u16 -> e5m2 code of the signed max-magnitude element of 8 N(0, sigma^2) samples (tail: _icdf_table)"""
    key = ("gf4scale", float(sigma), tail)
    if key not in _LUT_CACHE:
        rng = np.random.default_rng(12345)
        g = (rng.standard_normal((1 << 16, 8)) if tail == "normal" else rng.standard_t(4, size=(1 << 16, 8)) / math.sqrt(2.0)).astype(np.float32) * np.float32(sigma)
        idx = np.abs(g).argmax(-1)
        m = np.take_along_axis(g, idx[:, None], -1)[:, 0]
        _LUT_CACHE[key] = f32_to_fp8_e5m2(m.astype(np.float16).astype(np.float32))
    return _LUT_CACHE[key]


_SYNTH_LIB = None


def _synth_lib():
    """tools/libsynth_fill.so (OpenMP table walk); built on demand with gcc, None if that fails"""
    global _SYNTH_LIB
    if _SYNTH_LIB is None:
        import ctypes
        import os
        import subprocess

        tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
        so, src = os.path.join(tools, "libsynth_fill.so"), os.path.join(tools, "synth_fill.c")
        try:
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
                subprocess.run(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-o", so, src], check=True, capture_output=True)
            lib = ctypes.CDLL(so)
            lib.synth_fill.restype = None
            lib.synth_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
            _SYNTH_LIB = lib
        except Exception:
            _SYNTH_LIB = False
    return _SYNTH_LIB or None


def big_zeros(shape, dtype) -> np.ndarray:
    """np.zeros for the multi-GB host copies of synthetic models: anonymous memory with MADV_HUGEPAGE.  First touch of fresh guest
    memory runs at ~0.2 GB/s with 4 KiB pages inside these VMs and at ~9 GB/s with transparent huge pages (measured), which is
    what makes a 46.7 GB (Mixtral-8x7B) or 131.6 GB (DBRX-132B) model on the host a matter of seconds.  Small arrays and platforms
    without mmap.madvise take the plain path."""
    import mmap

    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes < (8 << 20) or not hasattr(mmap, "MADV_HUGEPAGE"):
        return np.zeros(shape, dtype=dtype)
    size = (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1)
    m = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    try:
        m.madvise(mmap.MADV_HUGEPAGE)
    except OSError:
        pass
    return np.frombuffer(m, dtype=dtype, count=int(np.prod(shape))).reshape(shape)  # (the array keeps the mapping alive)


def _fill_codes(out: np.ndarray, dtype: str, sigma: float, seed: int, tail: str = "normal") -> None:
    """fill the storage array `out` (uint16 / uint8 / uint32) with random weight codes"""
    flat = out.reshape(-1)
    lut = _gf4_scale_lut(sigma, tail) if dtype == "gf4" else _code_lut(dtype, sigma, tail)
    lib = _synth_lib()
    if lib is not None:
        kind = {"fp8": 0, "fp16": 1, "gf4": 2}[dtype]
        lib.synth_fill(flat.ctypes.data, flat.size, kind, lut.ctypes.data, seed)
        return
    rng = np.random.default_rng(seed)  # slow pure-numpy fallback, same distributions
    step = 1 << 22
    for a in range(0, flat.size, step):
        b = min(flat.size, a + step)
        u = rng.integers(0, 1 << 16, size=b - a, dtype=np.uint16)
        if dtype == "gf4":
            r = rng.integers(0, 1 << 32, size=b - a, dtype=np.uint32)
            k = (r & 7) * 3 + 8
            flat[a:b] = ((r & np.uint32(0xFFFFFF00)) & ~(np.uint32(7) << k)) | lut[u].astype(np.uint32)
        else:
            flat[a:b] = lut[u]


def _synth_walk(spec: "ModelSpec", dtype: str, seed: int, n_layers: Optional[int], W, F, with_tokenizer: bool = True, lognormal_norms: bool = False):
    """the tensor sequence of a synthetic model in file order: W(shape, sigma, fill_seed) makes a weight tensor, F(array) passes
    a small fp32 one on.  One walk serves the host filler (synth_stream_big) and the device filler (synth_device): same names,
    same order, same per-tensor seeds -- the two produce the same bytes."""
    s = dataclasses.replace(spec, n_layers=n_layers if n_layers is not None else spec.n_layers)
    E = s.n_experts
    lead = (E,) if E else ()
    rng = np.random.default_rng(seed)
    counter = [0]

    def WW(shape, fan_in, scale=1.0):
        counter[0] += 1
        return W(shape, scale / math.sqrt(fan_in), seed * 100003 + counter[0])

    def norm_w():
        if lognormal_norms:  # (synth_model_big(outliers=True): norm weights of trained models spread over an order of magnitude)
            return F(np.exp(0.4 * rng.standard_normal(s.dim)).astype(np.float32))
        return F((1 + 0.1 * rng.standard_normal(s.dim)).astype(np.float32))

    yield "model.embed.weight", WW((s.vocab_size, s.dim), 1.0)
    for l in range(s.n_layers):
        p = f"model.layers.{l}."
        yield p + "attn.norm.weight", norm_w()
        yield p + "attn.wq.weight", WW((s.q_dim, s.dim), s.dim)
        yield p + "attn.wk.weight", WW((s.kv_dim, s.dim), s.dim)
        yield p + "attn.wv.weight", WW((s.kv_dim, s.dim), s.dim)
        yield p + "attn.wo.weight", WW((s.dim, s.q_dim), s.q_dim)
        if s.norm_type != "layernorm_par":
            yield p + "mlp.norm.weight", norm_w()
        if E:
            yield p + "moegate.weight", WW((E, s.dim), s.dim, scale=4.0)
        yield p + "mlp.w1.weight", WW(lead + (s.hidden_dim, s.dim), s.dim)
        yield p + "mlp.w2.weight", WW(lead + (s.dim, s.hidden_dim), s.hidden_dim)
        yield p + "mlp.w3.weight", WW(lead + (s.hidden_dim, s.dim), s.dim)
    yield "model.norm.weight", norm_w()
    if not s.tied:
        yield "model.output.weight", WW((s.vocab_size, s.dim), s.dim)
    if with_tokenizer:
        toks, scores = _toy_tokenizer(s.vocab_size)
        yield "tokenizer.tokens", toks
        yield "tokenizer.scores", scores


def synth_stream_big(spec: ModelSpec, dtype: str, seed: int = 0, n_layers: Optional[int] = None, reuse: bool = True, tail: str = "normal"):
    """Yield (name, array) for every tensor of a full-size synthetic model, in file order, in seconds.

    Weight CODES are sampled directly -- fp16 / fp8: the quantised value of a N(0, 1/fan_in) draw
    (inverse-CDF table over 16 random bits); gf4: per word a scale drawn from the distribution of the
    signed max of 8 such draws, one code forced to 0 (the max element, as the real quantiser produces)
    and 7 uniform 3-bit codes.  Every tensor of every layer has its own random stream (seeded by name
    order), so the content does not depend on `reuse`.

    reuse=True: tensors of equal shape share ONE host buffer that is overwritten by the next tensor of
    that shape -- consume (upload) each array before advancing.  First touch of fresh guest memory
    runs at ~90 MB/s inside these VMs, so a 7 GB model held on the host costs 80 s, streamed 3 s.
    n_layers overrides spec.n_layers (layer-reduced models for CPU-affordable parity runs).
    """
    pool: Dict[Tuple, np.ndarray] = {}

    def W(shape, sigma, fill_seed):
        store = {"fp16": np.uint16, "fp8": np.uint8, "gf4": np.uint32}[dtype]
        sshape = shape if dtype != "gf4" else shape[:-1] + (shape[-1] // 8,)
        key = (sshape, store)
        a = pool.get(key) if reuse else None
        if a is None:
            a = big_zeros(sshape, store)
            pool[key] = a
        _fill_codes(a, dtype, sigma, fill_seed, tail)
        return a.view(np.float16) if dtype == "fp16" else (as_fp8(a) if dtype == "fp8" else a.view(np.int32))

    return _synth_walk(spec, dtype, seed, n_layers, W, lambda a: a, lognormal_norms=tail != "normal")


_DEV_SYNTH_LIB = None


def _dev_synth_lib():
    """tools/libsynth_fill_hip.so: the device-side filler / gf4 quantiser (fixture tooling; hipcc --offload-arch=gfx950)"""
    global _DEV_SYNTH_LIB
    if _DEV_SYNTH_LIB is None:
        import ctypes
        import os
        import subprocess

        tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
        so, src = os.path.join(tools, "libsynth_fill_hip.so"), os.path.join(tools, "synth_fill_hip.hip")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src],
                           check=True, capture_output=True)
        lib = ctypes.CDLL(so)
        lib.synth_fill_hip.restype = None
        lib.synth_fill_hip.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
        lib.quantize_gf4_hip.restype = None
        lib.quantize_gf4_hip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        _DEV_SYNTH_LIB = lib
    return _DEV_SYNTH_LIB


def synth_device(spec: ModelSpec, dtype: str, seed: int, n_layers: Optional[int], alloc, upload, before=None):
    """Yield (name, DEVICE pointer) for every `model.*` tensor of the synthetic model synth_stream_big describes -- the same
    bytes, produced on the GPU (tools/synth_fill_hip.hip): Mixtral-8x7B fp8 (46.7 GB) and DBRX-132B fp8 (131.6 GB) appear in
    HBM in seconds.  alloc(nbytes) -> device pointer with the backend's slack (alloc_hip); upload(array) -> device pointer
    (upload_hip; used for the small fp32 tensors and the code tables)."""
    lib = _dev_synth_lib()
    luts: Dict[Tuple, int] = {}
    kind = {"fp8": 0, "fp16": 1, "gf4": 2}[dtype]
    esize = {"fp8": 1, "fp16": 2, "gf4": 4}[dtype]

    where = [None]  # what before() returned for the tensor being made: the device it lives on (code tables are per device)

    def W(shape, sigma, fill_seed):
        n = int(np.prod(shape)) // (8 if dtype == "gf4" else 1)
        key = (dtype, float(sigma), where[0])
        if key not in luts:
            luts[key] = upload(np.ascontiguousarray(_gf4_scale_lut(sigma) if dtype == "gf4" else _code_lut(dtype, sigma)))
        ptr = alloc(n * esize)
        lib.synth_fill_hip(ptr, n, kind, luts[key], fill_seed)
        return ptr

    # `before(name)` runs ahead of each tensor's allocation (a multi-device host points the backend at the tensor's stage): the
    # walk is lazy -- a tensor is made when the loop below asks for it -- so the names are drawn from a dry walk first
    names = [n for n, _ in _synth_walk(spec, dtype, seed, n_layers, lambda *a: None, lambda a: None, with_tokenizer=False)]
    it = _synth_walk(spec, dtype, seed, n_layers, W, lambda a: upload(np.ascontiguousarray(a)), with_tokenizer=False)
    for nm in names:
        if before is not None:
            where[0] = before(nm)
        name, v = next(it)
        assert name == nm
        yield name, v
    if before is not None:
        before("")
    for ptr in luts.values():  # handed back so that the caller can free them with the tensors
        yield "", ptr


def stub_tensors(spec: ModelSpec, dtype: str, n_layers: Optional[int] = None) -> Dict[str, np.ndarray]:
    """name -> zero-stride placeholder with the right shape / dtype / nbytes and no memory behind it:
    enough for HostModel's config, accounting and pointer wiring when the real bytes are streamed"""
    s = dataclasses.replace(spec, n_layers=n_layers if n_layers is not None else spec.n_layers)
    E = s.n_experts
    lead = (E,) if E else ()
    store = {"fp16": np.float16, "fp8": np.uint8, "gf4": np.int32}[dtype]
    g = 8 if dtype == "gf4" else 1

    def W(*shape):
        a = np.broadcast_to(np.zeros(1, dtype=store), shape[:-1] + (shape[-1] // g,))
        return a.view(Fp8Bytes) if dtype == "fp8" else a

    def F(n):
        return np.broadcast_to(np.zeros(1, dtype=np.float32), (n,))

    t: Dict[str, np.ndarray] = {"model.embed.weight": W(s.vocab_size, s.dim)}
    for l in range(s.n_layers):
        p = f"model.layers.{l}."
        t[p + "attn.norm.weight"] = F(s.dim)
        t[p + "attn.wq.weight"] = W(s.q_dim, s.dim)
        t[p + "attn.wk.weight"] = W(s.kv_dim, s.dim)
        t[p + "attn.wv.weight"] = W(s.kv_dim, s.dim)
        t[p + "attn.wo.weight"] = W(s.dim, s.q_dim)
        if s.norm_type != "layernorm_par":
            t[p + "mlp.norm.weight"] = F(s.dim)
        if E:
            t[p + "moegate.weight"] = W(E, s.dim)
        t[p + "mlp.w1.weight"] = W(*lead, s.hidden_dim, s.dim)
        t[p + "mlp.w2.weight"] = W(*lead, s.dim, s.hidden_dim)
        t[p + "mlp.w3.weight"] = W(*lead, s.hidden_dim, s.dim)
    t["model.norm.weight"] = F(s.dim)
    if not s.tied:
        t["model.output.weight"] = W(s.vocab_size, s.dim)
    return t


OUTLIER_CHANNELS = 6


def outlier_channels(spec: ModelSpec, seed: int) -> np.ndarray:
    """the residual channels synth_model_big(outliers=True) inflates (sorted)"""
    return np.sort(np.random.default_rng(seed * 7919 + 13).choice(spec.dim, OUTLIER_CHANNELS, replace=False))


def _rescale(a: np.ndarray, dtype: str, rows, cols, factor: float) -> None:
    """multiply the weights a[rows, cols] (storage array of `dtype`; rows / cols: index arrays or None = all) by `factor`, in place,
    through a decode / re-quantise of the touched rows (gf4: of the touched 8-weight groups -- their other members keep their values
    up to the coarser code grid of the group's new scale)"""
    if dtype == "gf4":
        if cols is None:
            a[rows] = quantize_gf4(gf4_to_f32(a[rows]) * np.float32(factor))
            return
        for c in cols:  # one group column at a time (vocab x 8 floats)
            g = gf4_to_f32(a[:, c // 8:c // 8 + 1])
            g[:, c % 8] *= np.float32(factor)
            a[:, c // 8:c // 8 + 1] = quantize_gf4(g)
        return
    view = a.view(np.float16) if dtype == "fp16" else a.view(np.uint8)
    sel = (slice(None) if rows is None else rows, slice(None) if cols is None else cols)
    w = dequantize(view[sel], dtype) * np.float32(factor)
    view[sel] = quantize(w, dtype).view(view.dtype)


def synth_model_big(spec: ModelSpec, dtype: str, seed: int = 0, n_layers: Optional[int] = None, outliers: bool = False):
    """(tensors, metadata) of a full-shape synthetic model held entirely on the host (every tensor its
    own array) -- use for layer-reduced models; for 7 GB+ models stream with synth_stream_big.

    outliers=True (round 6): the statistics the default fixture lacks and trained checkpoints have.  N(0, 1 / fan_in) weights with norm
    weights 1 +- 0.1 keep every activation O(1) at any depth; a trained model's residual stream carries a handful of channels of
    10^2 - 10^3 ("massive activations"), its weights are heavy-tailed and its norm weights spread over an order of magnitude.  Here:
    Student-t (4 degrees of freedom, unit variance) weight entries, log-normal norm weights, and OUTLIER_CHANNELS residual channels
    whose embedding columns are scaled by 24 and whose wo / w2 output rows by 16 in every layer, so the channel random-walks to a few
    hundred by mid-depth while the others stay O(1) (dense models; tests/test_calmfile.py measures it with the oracle).
    outliers=2: in addition the w1 / w3 columns those channels feed are scaled by 4 / 4096, so that gated hidden activations pass 65504 --
    beyond binary16, where prefill_hip's hi + lo split gives up and sends the chunk back through the serial path (calm_hip.h:
    "pf_redone").  (The gate's argument has to stay below 88: the reference is built with -ffast-math and its SiLU, x / (1 + expf(-x)),
    turns a whole step into NaNs once expf overflows -- 128 on both columns did that at position 0; our restatement, built without
    fast-math, did not.)"""
    s = dataclasses.replace(spec, n_layers=n_layers if n_layers is not None else spec.n_layers)
    tensors = dict(synth_stream_big(spec, dtype, seed, n_layers, reuse=False, tail="t4" if outliers else "normal"))
    if outliers:
        assert not s.n_experts, "outliers: dense models"
        ch = outlier_channels(spec, seed)
        _rescale(tensors["model.embed.weight"], dtype, None, ch, 24.0)
        for l in range(s.n_layers):
            _rescale(tensors[f"model.layers.{l}.attn.wo.weight"], dtype, ch, None, 16.0)
            _rescale(tensors[f"model.layers.{l}.mlp.w2.weight"], dtype, ch, None, 16.0)
            if int(outliers) >= 2:
                _rescale(tensors[f"model.layers.{l}.mlp.w1.weight"], dtype, None, ch, 4.0)
                _rescale(tensors[f"model.layers.{l}.mlp.w3.weight"], dtype, None, ch, 4096.0)
    return tensors, s.metadata(dtype)


def write_synth_big(path: str, spec: ModelSpec, dtype: str, seed: int = 0, n_layers: Optional[int] = None) -> int:
    """a full-size synthetic .calm file (synth_stream_big's content) written with one tensor in memory at a time:
    the file the reference CLI -- or the CLI linked to libcalm_hip.so, INTEGRATION.md section A -- can be pointed at"""
    s = dataclasses.replace(spec, n_layers=n_layers if n_layers is not None else spec.n_layers)
    layout = stub_tensors(spec, dtype, n_layers)
    toks, scores = _toy_tokenizer(s.vocab_size)
    layout["tokenizer.tokens"] = toks
    layout["tokenizer.scores"] = scores
    return write_calm_stream(path, layout, synth_stream_big(spec, dtype, seed, n_layers, reuse=True), s.metadata(dtype))


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="write a synthetic .calm model of a BASELINE shape (seeded random weights, toy tokenizer)")
    ap.add_argument("model", choices=sorted(SPECS))
    ap.add_argument("dtype", choices=sorted(DBITS))
    ap.add_argument("out")
    ap.add_argument("--layers", type=int, default=None, help="cut the depth (a layer-reduced model for CPU-affordable runs)")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    n = write_synth_big(a.out, SPECS[a.model], a.dtype, a.seed, a.layers)
    print(f"{a.out}: {n / 1e9:.3f} GB, {a.model} {a.dtype}, {a.layers or SPECS[a.model].n_layers} layers")
