#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

The reference ships no golden vectors for the decode path, so parity is pinned
against outputs of the reference's own CPU backend: oracle/_ref/libcalm_ref.so is the untouched
/root/reference/src/infer.c (built by `make -C oracle ref`).  This script can therefore only run
where /root/reference exists (the build container); its outputs are committed:

    <case>.calm   the synthetic model (real file format, tiny shape, seeded weights)
    <case>.npz    tokens[T]  -- greedy token stream of the reference (teacher-forcing input)
                  logits[T, vocab] -- the reference's logits at every step
                  k_last / v_last  -- layer-0 KV cache rows after the last step (fp16 bits)
    cli_tiny_fp16.txt -- text decoded by the reference CLI (run_cpu) on one case
    cli_tiny_fp16_perplexity.txt -- its perplexity line for sample.txt (-x mode)

Usage:  python tests/golden/make_golden.py [case ...]
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from calm_amd import calmfile as cf  # noqa: E402
from calm_amd.host import HostModel  # noqa: E402
from oracle import oracle  # noqa: E402

# name -> (spec kwargs, dtype, steps, first token)
CASES = {
    "tiny_fp16": (dict(), "fp16", 24),
    "tiny_fp8": (dict(), "fp8", 24),
    "tiny_gf4": (dict(), "gf4", 24),
    "moe_fp8": (dict(n_experts=4, n_experts_active=2), "fp8", 24),
    "ln_gelu_clip_fp16": (dict(norm_type="layernorm", qkv_clip=2.0, act_type="gelu"), "fp16", 24),
    "par_fp8": (dict(norm_type="layernorm_par"), "fp8", 24),
    "bias_tied_gf4": (dict(qkv_bias=True, tied=True), "gf4", 24),
    # pos runs past seq_len: rolling buffer + attention-sink re-rotation (src/infer.c:329-332,383-394)
    "sink_fp16": (dict(max_seq_len=16), "fp16", 40),
    # ragged shapes: rows that are not a whole number of 1-KiB wave-loads, head_dim 32, kv_mul 1, odd vocab
    "ragged_fp8": (dict(dim=96, hidden_dim=176, head_dim=32, n_heads=3, n_kv_heads=3, vocab_size=301), "fp8", 16),
    # partial rotary embedding (phi-style): pairs beyond rotary_dim are not rotated (src/infer.c:226-228)
    "partial_rope_fp16": (dict(rotary_dim=8), "fp16", 24),
    # the DBRX combination at toy size: 16 experts top-4, LayerNorm without bias, qkv clip, kv_mul 3
    "dbrx_like_fp8": (dict(dim=96, hidden_dim=128, head_dim=16, n_heads=6, n_kv_heads=2, n_experts=16, n_experts_active=4, norm_type="layernorm", qkv_clip=1.5), "fp8", 24),
    # multi-query attention (one kv head), head_dim 96 (not a power of two: phi-3).  q_dim must not exceed dim in
    # a golden, nor dim hidden_dim: the reference's attention output lands in xb2 (dim floats) and wo's result in hb
    # (hidden_dim floats), src/infer.c:152-153,404,410, and both overflow otherwise
    "mqa_hd96_fp16": (dict(dim=192, hidden_dim=224, head_dim=96, n_heads=2, n_kv_heads=1, vocab_size=160), "fp16", 24),
    # mixture of experts over 4-bit weights, tied classifier
    "moe_gf4": (dict(n_experts=4, n_experts_active=2, tied=True), "gf4", 24),
    # expert counts that are NOT powers of two (round 4's routing-ahead pads its expert rows to a power of two and once read the norm
    # statistics from the padded rows: advisor finding): 6 experts top-2 under RMSNorm, 12 experts top-3 under LayerNorm
    "moe6_fp8": (dict(n_experts=6, n_experts_active=2), "fp8", 24),
    "moe12_ln_fp16": (dict(n_experts=12, n_experts_active=3, norm_type="layernorm"), "fp16", 24),
    # head_dim 256 (gemma-style), GELU, runs past a 12-row rolling buffer with fp8 weights
    "hd256_sink_fp8": (dict(dim=512, hidden_dim=544, head_dim=256, n_heads=2, n_kv_heads=1, n_layers=1, vocab_size=160, act_type="gelu", max_seq_len=12), "fp8", 30),
    # Shapes the fused k_qkv_attn launch takes (round 6: heads of 64 / 128, a 512-column input vector = whole-KiB fp16 rows; the tiny
    # cases above have heads of 16 and ragged rows and stay with k_qkv + k_attn): head 64 with q/k/v biases and two layers (the
    # hand-off granules' tags name the layer), head 128 past a 16-row rolling buffer (the masked slot is NOT the last row, sink keys)
    "fuse_hd64_bias_fp16": (dict(dim=512, hidden_dim=512, head_dim=64, n_heads=4, n_kv_heads=2, vocab_size=160, qkv_bias=True), "fp16", 24),
    "fuse_hd128_sink_fp16": (dict(dim=512, hidden_dim=512, head_dim=128, n_heads=4, n_kv_heads=2, n_layers=1, vocab_size=160, max_seq_len=16), "fp16", 40),
}
FIRST_TOKEN = 5


def main():
    if not oracle.have_ref():
        sys.exit("oracle/_ref/libcalm_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    only = set(sys.argv[1:])  # optional: regenerate just these cases (existing fixtures stay byte-identical)
    for name, (kw, dtype, steps) in CASES.items():
        if only and name not in only:
            continue
        spec = cf.tiny_spec(name, **kw)
        path = os.path.join(HERE, name + ".calm")
        cf.write_synth(path, spec, dtype, seed=1234)
        model = HostModel.from_file(path)
        ref = oracle.RefBackend(model)
        toks, logits = [], []
        tok = FIRST_TOKEN
        for pos in range(steps):
            lg = ref.forward(tok, pos, 0).copy()
            toks.append(tok)
            logits.append(lg)
            tok = int(np.argmax(lg))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            tokens=np.array(toks, dtype=np.int32),
            logits=np.stack(logits).astype(np.float32),
            k_last=ref.kv(0, 0).view(np.uint16).copy(),
            v_last=ref.kv(0, 1).view(np.uint16).copy(),
        )
        print(f"{name}: {os.path.getsize(path)} B model, {steps} steps, |logit|max {np.abs(logits[-1]).max():.3g}")

    if only and "cli" not in only:
        return
    # the reference CLI end to end (tokenizer + sampler + generate loop) on one case
    env = dict(os.environ, CALM_CPU="1", OMP_NUM_THREADS="2")
    r = subprocess.run([oracle.RUN_CPU, os.path.join(HERE, "tiny_fp16.calm"), "-i", "abc abc", "-t", "0", "-n", "32"], env=env, capture_output=True, text=True, check=True)
    with open(os.path.join(HERE, "cli_tiny_fp16.txt"), "w") as f:
        f.write(r.stdout.splitlines()[1] + "\n")  # line 0 is the model banner (path dependent), line 1 the decoded text
    print("cli:", r.stdout.splitlines()[1][:80])

    # perplexity mode (study(), src/run.c:258-316): positions wrap (pos = i % steps), every step wants logits
    r = subprocess.run([oracle.RUN_CPU, os.path.join(HERE, "tiny_fp16.calm"), "-x", os.path.join(HERE, "sample.txt"), "-n", "48"], env=env, capture_output=True, text=True, check=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("# perplexity:")][0]
    with open(os.path.join(HERE, "cli_tiny_fp16_perplexity.txt"), "w") as f:
        f.write(line.split("(")[0].strip() + "\n")
    print("pplx:", line)


if __name__ == "__main__":
    main()
