#!/bin/bash
OUT=gpurun_out/r03_ae; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== split attention over the transposed value cache: parity" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_long_context.py -m gpu -q -x -k "attention or long or context or golden or prefill or greedy" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -15 $OUT/pytest.log >> $OUT/summary.txt
echo "-- k_attn_vt" >> $OUT/summary.txt
timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
echo "-- k_attn_gqa (attn_vt = 0)" >> $OUT/summary.txt
CALM_HIP_ATTN_VT=0 timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
