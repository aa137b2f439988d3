#!/bin/bash
# First GPU-box session of round 2: the deep-prefetch persistent experiment (tools/exp_overlap.hip: k_deep, written at
# the end of round 1 and never run), then the round's usual validation.  ~1.5 GPU-minutes.
TAG=${1:-r02_open}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== deep-prefetch persistent chain vs launch per kernel (checksums must match)" | tee $OUT/summary.txt
timeout 180 python tools/exp_overlap.py deep > $OUT/exp_deep.txt 2>&1; echo "exit $?" >> $OUT/summary.txt
cat $OUT/exp_deep.txt >> $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "exit $?" >> $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log >> $OUT/summary.txt
cat $OUT/summary.txt
