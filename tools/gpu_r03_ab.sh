#!/bin/bash
OUT=gpurun_out/r03_ab; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity (norm scale left to the epilogue)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "decode_exact or matvec or golden or full_width or column_ranges or greedy or alternative" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log >> $OUT/summary.txt
for rep in 1 2; do
for cfg in "mistral-7b fp8 8" "llama-3-8b gf4 8" "tinyllama-1.1b fp16 22"; do
  echo "-- $cfg" >> $OUT/summary.txt
  timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
