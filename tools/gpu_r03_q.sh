#!/bin/bash
OUT=gpurun_out/r03_q; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity (new tile depths)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or full_width or matvec or norm" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
for rep in 1 2; do
for lib in libcalm_hip.so libcalm_hip_old.so libcalm_hip_ao2.so libcalm_hip_out4.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
done
for cfg in "tinyllama-1.1b fp16 22" "mixtral-8x7b fp8 4" "dbrx-132b fp8 2" "mistral-7b fp16 4"; do
for lib in libcalm_hip.so libcalm_hip_old.so; do
  echo "-- $lib $cfg" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
