#!/bin/bash
OUT=gpurun_out/r03_x; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity (shape rules, one row per task)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "decode_exact or matvec or golden or full_width or column_ranges or greedy or alternative" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log >> $OUT/summary.txt
for lib in libcalm_hip.so libcalm_hip_one1.so; do
for cfg in "dbrx-132b fp8 2" "tinyllama-1.1b fp16 22"; do
  echo "-- $lib $cfg" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
done
echo "-- dbrx rules off" >> $OUT/summary.txt
KNOBS="out_one=2 down_one=2" timeout 300 python tools/tune.py dbrx-132b fp8 2 brief >> $OUT/summary.txt 2>&1
echo "-- mixtral" >> $OUT/summary.txt
timeout 300 python tools/tune.py mixtral-8x7b fp8 4 brief >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
