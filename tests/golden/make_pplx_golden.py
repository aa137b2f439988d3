#!/usr/bin/env python3
"""Golden for the reference's perplexity mode on ITS OWN text at a BASELINE width.

    `run -x tools/pplx.txt` (study(), src/run.c:258-316) on the reference's CPU backend, over a 4-layer Mistral-7B-width fp8
    model (dim 4096, hidden 14336, vocabulary 32000; seeded synthetic weights, toy tokenizer), positions wrapping every 1024.

The text is the reference's tools/pplx.txt.  It is stored here as the byte values the toy tokenizer maps one-to-one to token ids
(pplx_bytes.npz) next to the perplexity line the reference printed for it (cli_mistral4_pplx_perplexity.txt); the GPU test
writes the bytes back to a text file, rebuilds the same model from its seed, and runs the unmodified reference CLI on the HIP
backend.  Runs only where /root/reference exists (about 20 minutes of CPU on 8 cores).

Usage:  python tests/golden/make_pplx_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from calm_amd import calmfile as cf  # noqa: E402
from oracle import oracle  # noqa: E402

MODEL, DTYPE, LAYERS, SEED, CHUNK = "mistral-7b", "fp8", 4, 5, 1024

if __name__ == "__main__":
    text = open("/root/reference/tools/pplx.txt", "rb").read()
    np.savez_compressed(os.path.join(HERE, "pplx_bytes.npz"), bytes=np.frombuffer(text, dtype=np.uint8))
    with tempfile.TemporaryDirectory() as tmp:
        model = os.path.join(tmp, "m.calm")
        cf.write_synth_big(model, cf.SPECS[MODEL], DTYPE, SEED, LAYERS)
        txt = os.path.join(tmp, "pplx.txt")
        open(txt, "wb").write(text)
        env = dict(os.environ, CALM_CPU="1")
        r = subprocess.run([oracle.RUN_CPU, model, "-x", txt, "-n", str(CHUNK)], env=env, capture_output=True, text=True, check=True)
    head = [l for l in r.stdout.splitlines() if "tokens (" in l][0]
    line = [l for l in r.stdout.splitlines() if l.startswith("# perplexity:")][0]
    with open(os.path.join(HERE, "cli_mistral4_pplx_perplexity.txt"), "w") as f:
        f.write(head.split(":", 1)[1].strip() + "\n" + line + "\n")
    print(head)
    print(line)
