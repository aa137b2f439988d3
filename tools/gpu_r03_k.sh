#!/bin/bash
OUT=gpurun_out/r03_k; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
timeout 300 python tools/loadprobe.py 2>&1 | tee $OUT/summary.txt
echo "== gf4 grid caps" | tee -a $OUT/summary.txt
KNOB_VALUES=1,2,3 timeout 300 python tools/ab_knob.py bpc llama-3-8b gf4 8 2>&1 | tee -a $OUT/summary.txt
echo "== tinyllama fp16 grid caps" | tee -a $OUT/summary.txt
KNOB_VALUES=1,2,3 timeout 300 python tools/ab_knob.py bpc tinyllama-1.1b fp16 22 2>&1 | tee -a $OUT/summary.txt
