#!/bin/bash
OUT=gpurun_out/r03_n; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== MoE parity" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_multi_device.py -m gpu -q -x -k "moe or dbrx or golden" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
for cfg in "mixtral-8x7b fp8 4" "dbrx-132b fp8 2"; do
  timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
