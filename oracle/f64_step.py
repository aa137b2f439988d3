"""float64 evaluation of the FIRST decode step of a dense model (test infrastructure, like everything under oracle/).

With one cached position the attention output is the value row itself, so the step is embedding -> per layer [norm, wv, wo, norm, w1 /
w3 / w2] -> norm -> classifier (src/infer.c:311-472 with kv_len = 1; the value row rounded to binary16 as the cache keeps it,
src/infer.c:378-381).  Used to tell which of two fp32 implementations that disagree at the 1e-3 level is further from exact arithmetic
(tools/outlier_f64.py, tests/test_full_depth_parity.py); RMSNorm models without biases, clipping or experts."""
import numpy as np

from calm_amd import calmfile as cf


def position0_logits_f64(tensors, md, spec, dtype, tokens):
    """-> float64 logits [len(tokens)][vocab] of decoding each of `tokens` at position 0"""
    assert not spec.n_experts and spec.norm_type == "rmsnorm" and not spec.qkv_bias and spec.qkv_clip is None
    D = lambda n: cf.dequantize(tensors[n], dtype).astype(np.float64)
    eps = float(md["norm_eps"])
    rms = lambda X, g: X / np.sqrt((X * X).mean(0, keepdims=True) + eps) * g.astype(np.float64)[:, None]
    kvm = spec.n_heads // spec.n_kv_heads
    n_layers = int(md["n_layers"])
    X = np.stack([cf.dequantize(tensors["model.embed.weight"][t:t + 1], dtype).astype(np.float64)[0] for t in tokens], 1)  # (dim, tokens)
    for l in range(n_layers):
        p = f"model.layers.{l}."
        v = (D(p + "attn.wv.weight") @ rms(X, tensors[p + "attn.norm.weight"])).astype(np.float16).astype(np.float64)
        att = np.concatenate([v[(h // kvm) * spec.head_dim:(h // kvm + 1) * spec.head_dim] for h in range(spec.n_heads)])
        X = X + D(p + "attn.wo.weight") @ att
        xb = rms(X, tensors[p + "mlp.norm.weight"])
        a, b = D(p + "mlp.w1.weight") @ xb, D(p + "mlp.w3.weight") @ xb
        X = X + D(p + "mlp.w2.weight") @ (a / (1 + np.exp(-a)) * b)
    w = "model.embed.weight" if "model.output.weight" not in tensors else "model.output.weight"
    return (D(w) @ rms(X, tensors["model.norm.weight"])).T
