#!/bin/bash
# Round 3, third GPU session: the loader study behind the engine gate, the perplexity pin on the reference text, wave timelines.
TAG=${1:-r03_c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== loader study" | tee $OUT/summary.txt
timeout 600 python tools/exp_loader.py > $OUT/loader.txt 2>&1; echo "exit $?" >> $OUT/summary.txt
cat $OUT/loader.txt >> $OUT/summary.txt
echo "== perplexity pin" | tee -a $OUT/summary.txt
( time timeout 900 python -m pytest tests/test_hip_parity.py -q -x -s -k "perplexity" ) > $OUT/pytest_pplx.log 2>&1
echo "exit $?" >> $OUT/summary.txt; grep -E "perplexity of|passed|failed|Error|real|assert" $OUT/pytest_pplx.log | tail -8 >> $OUT/summary.txt
echo "== timeline" | tee -a $OUT/summary.txt
timeout 300 python tools/timeline.py >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
