#!/usr/bin/env python3
"""Round 5, verdict item 6 (iii), the error study that has to come BEFORE any kernel: could the `lo` half of the prompt GEMMs' hi + lo
activations ride an fp8 matrix-core product (v_mfma_scale_f32_32x32x64_f8f6f4: ONE instruction per 64 columns instead of two of the four
f16 MFMAs)?  numpy, no GPU:  python tools/lo_fp8_study.py

The product today: y = W.(hi + lo), hi = half(x), lo = half(x - hi); both products exact in fp32 (weights decode exactly to binary16).
The priced variant: W.hi as now (f16), W.lo with lo as BLOCK-SCALED fp8 (one power-of-two scale per 32 columns, the MX layout the scaled
MFMA takes) in e4m3 (3 mantissa bits) or e5m2 (2), and the weights as fp8 too: an fp8 (e5m2) .calm weight is exact there; a gf4 or fp16
weight would have to be ROUNDED to fp8 for this term (its error rides on the 2^-11 of lo).  Bar: max|err| / max|y|
<= 1e-6 per GEMM against a float64 dot product -- today's hi + lo form measures 3-7e-7.
"""
import numpy as np

rng = np.random.default_rng(0)


def to_fp8(v, mant, emin, emax_val):
    """round-to-nearest-even to a binary8 format with `mant` mantissa bits, smallest normal exponent emin, saturating at emax_val"""
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros_like(v)
    nz = v != 0
    e = np.floor(np.log2(np.abs(v[nz])))
    e = np.maximum(e, emin)  # subnormals share the smallest exponent
    q = np.round(v[nz] / 2.0 ** (e - mant)) * 2.0 ** (e - mant)  # (np.round is half-to-even)
    out[nz] = np.clip(q, -emax_val, emax_val)
    return out


def e4m3(v):
    return to_fp8(v, 3, -6, 448.0)


def e5m2(v):
    return to_fp8(v, 2, -14, 57344.0)


def block_scaled(lo, fmt, top):
    """MX-style: per 32 columns one power-of-two scale that puts the block's largest magnitude just inside the format's range"""
    lo = lo.astype(np.float64).reshape(-1, 32)
    amax = np.abs(lo).max(axis=1, keepdims=True)
    scale = 2.0 ** (np.floor(np.log2(np.maximum(amax, 1e-300))) - np.floor(np.log2(top)))
    return (fmt(lo / scale) * scale).reshape(-1)


def study(n, d, xscale, wfmt, outliers=0):
    W = (rng.standard_normal((d, n)) * n ** -0.5).astype(np.float32)
    if wfmt == "fp8":
        W = e5m2(W).astype(np.float32)  # an fp8 .calm weight: exact in the fp8 product
        Wlo = W.astype(np.float64)
    else:
        W = W.astype(np.float16).astype(np.float32)  # fp16 / gf4-decoded weights: exact in binary16, ROUNDED to e5m2 / e4m3 for the lo term
        Wlo = None
    x = (rng.standard_normal(n) * xscale).astype(np.float32)
    if outliers:
        x[rng.integers(0, n, outliers)] *= 200
    ref = W.astype(np.float64) @ x.astype(np.float64)
    sc = np.abs(ref).max()
    xh = x.astype(np.float16)
    lo = x - xh.astype(np.float32)
    xl = lo.astype(np.float16)
    W64 = W.astype(np.float64)
    base = W64 @ xh.astype(np.float64)
    res = {"hi+lo f16 (today)": base + W64 @ xl.astype(np.float64), "hi only": base}
    for name, fmt, top in (("lo e4m3/32", e4m3, 448.0), ("lo e5m2/32", e5m2, 57344.0)):
        l8 = block_scaled(lo, fmt, top)
        if Wlo is not None:
            res[name] = base + Wlo @ l8
        else:
            for wname, wf in (("W->e5m2", e5m2), ("W->e4m3", e4m3)):
                res[f"{name}, {wname}"] = base + wf(W64) @ l8
    return {k: float(np.abs(v - ref).max() / sc) for k, v in res.items()}


def main():
    print("max |err| / max |y| against float64, per GEMM (bar: <= 1e-6; today's form: 3-7e-7 measured on the GPU, exact-product arithmetic here)\n")
    for wfmt in ("fp8", "fp16"):
        for n, d, xs, o in ((4096, 4096, 1.0, 0), (4096, 4096, 1.0, 8), (14336, 4096, 1.0, 0), (4096, 14336, 30.0, 0)):
            r = study(n, d, xs, wfmt, o)
            print(f"weights {wfmt:4s} K {n:6d} M {d:6d} x-scale {xs:6.3g} outliers {o}: " + "  ".join(f"{k}: {v:.2e}" for k, v in r.items()))
        print()


if __name__ == "__main__":
    main()
