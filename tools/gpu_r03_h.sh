#!/bin/bash
OUT=gpurun_out/r03_h; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== attention parity (matrix-core split attention)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "attn or attention or long_context or split or golden" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -12 $OUT/pytest.log >> $OUT/summary.txt
for m in 0 1; do
echo "== long context, CALM_HIP_ATTN_MFMA=$m" | tee -a $OUT/summary.txt
CALM_HIP_ATTN_MFMA=$m timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
