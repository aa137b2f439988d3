#!/bin/bash
OUT=gpurun_out/r03_o; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== parity (ffn_down forms)" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or full_width" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
echo "== down_u (exact 2-chunk steps), Mistral-7B fp8" >> $OUT/summary.txt
KNOB_VALUES=0,2 timeout 300 python tools/ab_knob.py down_u mistral-7b fp8 8 >> $OUT/summary.txt 2>&1
echo "== down_next, Mixtral fp8 4 layers" >> $OUT/summary.txt
KNOB_VALUES=0,1 timeout 300 python tools/ab_knob.py down_next mixtral-8x7b fp8 4 >> $OUT/summary.txt 2>&1
echo "== down_u, Mixtral fp8 4 layers" >> $OUT/summary.txt
KNOB_VALUES=0,2 timeout 300 python tools/ab_knob.py down_u mixtral-8x7b fp8 4 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
