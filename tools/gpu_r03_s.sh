#!/bin/bash
OUT=gpurun_out/r03_s; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== gf4: tile depth x resident workgroups, old (multiply-add) and new (matrix-core) forms" | tee $OUT/summary.txt
for lib in libcalm_hip_old.so libcalm_hip_o41.so libcalm_hip_o22.so libcalm_hip_vf.so libcalm_hip_n4.so libcalm_hip_n2.so; do
  [ -f calm_amd/$lib ] || continue
  for bpc in 2 4; do
  echo "-- $lib BPC=$bpc" >> $OUT/summary.txt
  BPCS="2 3 4" BPC=$bpc CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py llama-3-8b gf4 8 brief >> $OUT/summary.txt 2>&1
  done
done
cat $OUT/summary.txt
