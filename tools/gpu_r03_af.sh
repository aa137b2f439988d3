#!/bin/bash
OUT=gpurun_out/r03_af; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== k_attn_vt: lazy rescale; split lengths" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_long_context.py -m gpu -q -x -k "attention or long or context" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -4 $OUT/pytest.log >> $OUT/summary.txt
SPLITS="0,64,256" timeout 600 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
