#!/bin/bash
# Short GPU-box session for kernel iteration: small parity tests, experiments, stage sweep, bench without the CPU leg.
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== kernel + tiny-model parity tests" | tee $OUT/summary.txt
timeout 600 python -m pytest tests -m gpu -q -k "not full_width and not full_depth" > $OUT/pytest_small.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -8 $OUT/pytest_small.log >> $OUT/summary.txt
if [ -f tools/exp_overlap.py ] && [ "${EXP:-1}" = "1" ]; then
  echo "== overlap experiment" | tee -a $OUT/summary.txt
  timeout 300 python tools/exp_overlap.py > $OUT/exp_overlap.txt 2>&1; cat $OUT/exp_overlap.txt >> $OUT/summary.txt
fi
echo "== tune" | tee -a $OUT/summary.txt
timeout 600 python tools/tune.py > $OUT/tune.txt 2>&1; cat $OUT/tune.txt >> $OUT/summary.txt
if [ "${SHAPES:-0}" = "1" ]; then
  echo "== other BASELINE shapes (layer-reduced)" | tee -a $OUT/summary.txt
  for cfg in "llama-3-8b gf4 6" "tinyllama-1.1b fp16 12" "mixtral-8x7b fp8 2" "dbrx-132b fp8 1" "mistral-7b fp16 4" "mistral-7b gf4 8"; do
    timeout 300 python tools/tune.py $cfg brief >> $OUT/shapes.txt 2>&1
  done
  cat $OUT/shapes.txt >> $OUT/summary.txt
fi
echo "== bench (no cpu leg)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/summary.txt
tail -c 2500 $OUT/bench.json >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt
cat $OUT/summary.txt
