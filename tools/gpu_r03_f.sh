#!/bin/bash
# Round 3 session F: attention changes -- parity, the unsplit kernel's wave count, long-context stage times
OUT=gpurun_out/r03_f; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== attention / golden / long-context parity" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "attn or attention or golden or long_context or split" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log >> $OUT/summary.txt
echo "== unsplit attention: waves per workgroup" | tee -a $OUT/summary.txt
KNOB_VALUES=16,8,4 timeout 300 python tools/ab_knob.py attn_waves mistral-7b fp8 8 >> $OUT/summary.txt 2>&1
echo "== long context" | tee -a $OUT/summary.txt
timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
