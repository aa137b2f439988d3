#!/bin/bash
# GPU-box session: the whole GPU test suite as the driver runs it, the smoke entry, the default bench.
TAG=${1:-validate}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== pytest -m gpu" | tee $OUT/summary.txt
( time timeout 900 python -m pytest tests -x -q -m gpu --durations=15 ) > $OUT/pytest_gpu.log 2>&1; echo "exit $?" >> $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 200 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/smoke.log >> $OUT/summary.txt
if [ "${BENCH:-1}" = "1" ]; then
  echo "== bench" | tee -a $OUT/summary.txt
  ( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/summary.txt
  tail -c 3000 $OUT/bench.json >> $OUT/summary.txt; tail -8 $OUT/bench.err >> $OUT/summary.txt
fi
cat $OUT/summary.txt
