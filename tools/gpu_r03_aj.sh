#!/bin/bash
OUT=gpurun_out/r03_aj; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== MoE gate rows beyond the prefetch asked for in batches" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "moe or dbrx or golden or full_width" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -4 $OUT/pytest.log >> $OUT/summary.txt
for rep in 1 2; do
for lib in libcalm_hip_prev.so libcalm_hip.so; do
  for cfg in "dbrx-132b fp8 2" "mixtral-8x7b fp8 4"; do
  echo "-- $lib $cfg" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
  done
done
done
cat $OUT/summary.txt
