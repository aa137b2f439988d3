#!/bin/bash
OUT=gpurun_out/r03_v; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== fp8 / fp16 tile shapes" | tee $OUT/summary.txt
for cfg in "mistral-7b fp8 8" "tinyllama-1.1b fp16 22"; do
for lib in libcalm_hip.so libcalm_hip_v1.so libcalm_hip_v2.so libcalm_hip_v3.so libcalm_hip_v4.so; do
  echo "-- $lib $cfg down_u=0" >> $OUT/summary.txt
  KNOBS="down_u=0" CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
echo "-- libcalm_hip.so $cfg (product)" >> $OUT/summary.txt
timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
