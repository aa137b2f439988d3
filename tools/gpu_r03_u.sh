#!/bin/bash
OUT=gpurun_out/r03_u; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity with per-kernel gf4 tile shapes" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "decode_exact or matvec or golden or full_width or column_ranges or greedy" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log >> $OUT/summary.txt
for rep in 1 2; do
for lib in libcalm_hip.so libcalm_hip_a.so libcalm_hip_b.so libcalm_hip_sw.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py llama-3-8b gf4 8 brief >> $OUT/summary.txt 2>&1
done
done
for lib in libcalm_hip.so libcalm_hip_sw.so libcalm_hip.so libcalm_hip_sw.so; do
  echo "-- $lib mistral fp8" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
