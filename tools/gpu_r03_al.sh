#!/bin/bash
OUT=gpurun_out/r03_al; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== k_attn: K rows of a round asked for ahead of the V rows" | tee $OUT/summary.txt
for rep in 1 2 3; do
for lib in libcalm_hip_kvint.so libcalm_hip_kfirst.so; do
  for cfg in "mistral-7b fp8 8" "tinyllama-1.1b fp16 22"; do
  echo "-- $lib $cfg" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
  done
done
done
cat $OUT/summary.txt
