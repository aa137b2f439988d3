#!/bin/bash
OUT=gpurun_out/r03_ai; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== prompt attention over the transposed value cache" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_long_context.py tests/test_full_depth_parity.py -m gpu -q -x -k "prefill or perplexity or long or context or full_depth" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -6 $OUT/pytest.log >> $OUT/summary.txt
for kn in "attn_vt=1" "attn_vt=0"; do
  echo "-- $kn" >> $OUT/summary.txt
  KNOBS="$kn" timeout 300 python tools/prefill_bench.py mistral-7b fp8 4 2048 >> $OUT/summary.txt 2>&1
  KNOBS="$kn" timeout 300 python tools/prefill_bench.py mistral-7b fp8 4 4000 2>&1 | grep "prefill  4000" >> $OUT/summary.txt
done
cat $OUT/summary.txt
