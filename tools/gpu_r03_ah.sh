#!/bin/bash
OUT=gpurun_out/r03_ah; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== probe: k_ffn_up's image by LDS-DMA (timing only)" | tee $OUT/summary.txt
for rep in 1 2 3; do
for lib in libcalm_hip.so libcalm_hip_dma.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
