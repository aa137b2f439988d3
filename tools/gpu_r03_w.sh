#!/bin/bash
OUT=gpurun_out/r03_w; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity (shape rules)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "decode_exact or matvec or golden or full_width or column_ranges or greedy" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log >> $OUT/summary.txt
for cfg in "tinyllama-1.1b fp16 22" "llama-3-8b gf4 8" "mistral-7b fp8 8" "mixtral-8x7b fp8 4" "dbrx-132b fp8 2"; do
  echo "-- $cfg" >> $OUT/summary.txt
  timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
echo "-- tinyllama rules off" >> $OUT/summary.txt
KNOBS="qkv_half=2 down_one=2" timeout 300 python tools/tune.py tinyllama-1.1b fp16 22 brief >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
