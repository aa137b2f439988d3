#!/bin/bash
OUT=gpurun_out/r03_t; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== gf4 tile shapes (rows x KiB per tile), multiply-add form" | tee $OUT/summary.txt
for lib in libcalm_hip_o21.so; do
  echo "-- $lib" >> $OUT/summary.txt
  BPCS="2 3 4" CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py llama-3-8b gf4 8 brief >> $OUT/summary.txt 2>&1
done
for lib in libcalm_hip_old.so libcalm_hip_o41.so libcalm_hip_o22.so libcalm_hip_o21.so; do
  echo "-- $lib down_u4=0" >> $OUT/summary.txt
  KNOBS="down_u4=0" BPCS="2 4" CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py llama-3-8b gf4 8 brief >> $OUT/summary.txt 2>&1
done
echo "== timeline (4 rows x 2 KiB tiles)" >> $OUT/summary.txt
timeout 300 python tools/timeline.py llama-3-8b gf4 8 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
