#!/bin/bash
OUT=gpurun_out/r03_z; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity (row pairs shared by two waves in k_qkv)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "decode_exact or matvec or golden or full_width or column_ranges or greedy or alternative" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log >> $OUT/summary.txt
for cfg in "mistral-7b fp8 8" "llama-3-8b gf4 8" "mixtral-8x7b fp8 4"; do
for kn in "qkv_mode=0" "qkv_mode=1" "qkv_mode=2" "qkv_mode=3"; do
  echo "-- $cfg $kn" >> $OUT/summary.txt
  KNOBS="$kn" timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
