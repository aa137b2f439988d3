"""ctypes mirror of include/calm_abi.h (== reference src/model.h:12-89).

The field order and types below ARE the drop-in contract: a `Transformer` built here can be
handed to libcalm_hip.so (forward_hip), to the oracle (oracle_forward) and to the reference's own
CPU backend (forward in src/infer.c) without conversion.  tests/test_abi.py pins every offset.
"""
import ctypes as C

MAX_LAYERS = 128
MAX_EXPERTS = 64
KV_SINKS = 2

FF_UPDATE_KV_ONLY = 1 << 0

_vp = C.c_void_p
_fp = C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [
        ("dim", C.c_int),
        ("hidden_dim", C.c_int),
        ("head_dim", C.c_int),
        ("n_layers", C.c_int),
        ("n_heads", C.c_int),
        ("n_kv_heads", C.c_int),
        ("vocab_size", C.c_int),
        ("seq_len", C.c_int),
        ("rope_theta", C.c_float),
        ("rotary_dim", C.c_int),
        ("n_experts", C.c_int),
        ("n_experts_ac", C.c_int),
        ("norm_eps", C.c_float),
        ("act_gelu", C.c_bool),
        ("norm_ln", C.c_bool),
        ("norm_par", C.c_bool),
        ("qkv_clip", C.c_float),
    ]


class Weights(C.Structure):
    _fields_ = [
        ("dbits", C.c_int),
        ("token_embedding_table", _vp),
        ("rms_att_weight", _vp * MAX_LAYERS),
        ("rms_ffn_weight", _vp * MAX_LAYERS),
        ("wq", _vp * MAX_LAYERS),
        ("wk", _vp * MAX_LAYERS),
        ("wv", _vp * MAX_LAYERS),
        ("wo", _vp * MAX_LAYERS),
        ("w1", _vp * MAX_LAYERS),
        ("w2", _vp * MAX_LAYERS),
        ("w3", _vp * MAX_LAYERS),
        ("rms_final_weight", _vp),
        ("wcls", _vp),
        ("bqkv", _vp * MAX_LAYERS),
        ("moegate", _vp * MAX_LAYERS),
    ]


class RunState(C.Structure):
    _fields_ = [
        ("x", _vp),
        ("xb", _vp),
        ("xb2", _vp),
        ("hb", _vp),
        ("hb2", _vp),
        ("he", _vp),
        ("q", _vp),
        ("k", _vp),
        ("v", _vp),
        ("att", _vp),
        ("exp", _vp),
        ("logits", _vp),
        ("kvbits", C.c_int),
        ("key_cache", _vp),
        ("value_cache", _vp),
    ]


class Transformer(C.Structure):
    pass


class Sampler(C.Structure):
    """reference src/sampler.h:3-9 (include/calm_abi.h)"""

    _fields_ = [("vocab_size", C.c_int), ("rng_state", C.c_ulonglong), ("temperature", C.c_float), ("minp", C.c_float)]


FORWARD_FN = C.CFUNCTYPE(_fp, C.POINTER(Transformer), C.c_int, C.c_int, C.c_uint)

Transformer._fields_ = [
    ("config", Config),
    ("weights", Weights),
    ("state", RunState),
    ("n_params", C.c_size_t),
    ("n_bytes", C.c_size_t),
    ("n_bandwidth", C.c_size_t),
    ("forward", _vp),
]

# frozen layout numbers (x86-64 / LP64); checked against the reference header in tests/test_abi.py
FROZEN_LAYOUT = {
    "sizeof(Config)": 60,
    "sizeof(Weights)": 11296,
    "sizeof(RunState)": 120,
    "sizeof(Transformer)": 11512,
    "Config.qkv_clip": 56,
    "Config.act_gelu": 52,
    "Weights.token_embedding_table": 8,
    "Weights.rms_final_weight": 9232,
    "Weights.wcls": 9240,
    "Weights.bqkv": 9248,
    "Weights.moegate": 10272,
    "RunState.logits": 88,
    "RunState.kvbits": 96,
    "RunState.key_cache": 104,
    "Transformer.weights": 64,
    "Transformer.state": 11360,
    "Transformer.n_params": 11480,
    "Transformer.n_bandwidth": 11496,
    "Transformer.forward": 11504,
}


def layout():
    """actual ctypes layout in the same keys as FROZEN_LAYOUT"""
    out = {
        "sizeof(Config)": C.sizeof(Config),
        "sizeof(Weights)": C.sizeof(Weights),
        "sizeof(RunState)": C.sizeof(RunState),
        "sizeof(Transformer)": C.sizeof(Transformer),
    }
    for cls in (Config, Weights, RunState, Transformer):
        for name, _ in cls._fields_:
            key = f"{cls.__name__}.{name}"
            if key in FROZEN_LAYOUT:
                out[key] = getattr(cls, name).offset
    return out
