#!/bin/bash
OUT=gpurun_out/r03_y; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== resident workgroups per CU, per stage" | tee $OUT/summary.txt
for cfg in "mistral-7b fp8 8" "dbrx-132b fp8 2" "tinyllama-1.1b fp16 22"; do
  echo "-- $cfg" >> $OUT/summary.txt
  BPCS="2 3 4" timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
