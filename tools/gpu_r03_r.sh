#!/bin/bash
OUT=gpurun_out/r03_r; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== gf4 on the matrix cores: parity" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "mfma_4x4x4 or decode_exact or matvec or golden or full_width or column_ranges or greedy" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -15 $OUT/pytest.log >> $OUT/summary.txt
for rep in 1 2; do
for lib in libcalm_hip.so libcalm_hip_old.so libcalm_hip_vf.so; do
  [ -f calm_amd/$lib ] || continue
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py llama-3-8b gf4 8 brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
