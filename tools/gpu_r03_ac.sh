#!/bin/bash
OUT=gpurun_out/r03_ac; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== A/B on one box: norm scale in the image (prev) / left to the epilogue" | tee $OUT/summary.txt
for rep in 1 2 3; do
for cfg in "mistral-7b fp8 8" "mixtral-8x7b fp8 4"; do
for lib in libcalm_hip_prev.so libcalm_hip.so; do
  echo "-- $lib $cfg" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
done
done
cat $OUT/summary.txt
