#!/bin/bash
OUT=gpurun_out/r03_p; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== pytest -m gpu (whole suite)" | tee $OUT/summary.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1; echo "exit $?" >> $OUT/summary.txt
tail -8 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== tile shapes" >> $OUT/summary.txt
for rep in 1 2; do
for lib in libcalm_hip.so libcalm_hip_u2.so libcalm_hip_nr4u2.so libcalm_hip_u1.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
