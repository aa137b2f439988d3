#!/bin/bash
# GPU-box session for the chained-launch experiment: any-order probe + synthetic chain, chained-step parity tests,
# A/B on an 8-layer model, the new golden cases, bench with launch-mode calibration.  Everything lands in gpurun_out/$TAG.
TAG=${1:-chain}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== any-order probe + synthetic chain" | tee $OUT/summary.txt
timeout 120 python tools/exp_overlap.py anyorder > $OUT/exp_anyorder.txt 2>&1; echo "exit $?" >> $OUT/summary.txt
cat $OUT/exp_anyorder.txt >> $OUT/summary.txt
echo "== chained-step parity tests" | tee -a $OUT/summary.txt
timeout 300 python -m pytest tests/test_zz_chained_launches.py -m gpu -q > $OUT/pytest_chain.log 2>&1; echo "exit $?" >> $OUT/summary.txt
tail -25 $OUT/pytest_chain.log >> $OUT/summary.txt
echo "== chain A/B, 8-layer mistral-7b fp8" | tee -a $OUT/summary.txt
timeout 150 python tools/chain_ab.py mistral-7b fp8 8 > $OUT/chain_ab.txt 2>&1; echo "exit $?" >> $OUT/summary.txt
cat $OUT/chain_ab.txt >> $OUT/summary.txt
echo "== new golden cases on the GPU" | tee -a $OUT/summary.txt
timeout 240 python -m pytest tests/test_hip_parity.py -m gpu -q -k "partial_rope or dbrx_like or mqa_hd96 or moe_gf4 or hd256" > $OUT/pytest_newgold.log 2>&1; echo "exit $?" >> $OUT/summary.txt
tail -25 $OUT/pytest_newgold.log >> $OUT/summary.txt
echo "== bench (launch mode calibrated, no cpu leg)" | tee -a $OUT/summary.txt
timeout 400 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/summary.txt
tail -c 3500 $OUT/bench.json >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt
cat $OUT/summary.txt
