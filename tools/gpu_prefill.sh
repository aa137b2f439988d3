#!/bin/bash
# GPU session for the prompt-ingestion path: its parity tests (with durations), the rate tool, a kernel trace
TAG=${1:-pf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -q -k "prefill and not full_width" --durations=8 > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -15 $OUT/pytest.log >> $OUT/summary.txt
for cfg in "mistral-7b fp8 4 512" "llama-3-8b gf4 4 256" "tinyllama-1.1b fp16 8 256"; do
  timeout 300 python tools/prefill_bench.py $cfg >> $OUT/prefill.txt 2>&1
done
cat $OUT/prefill.txt >> $OUT/summary.txt
if [ "${PROF:-1}" = "1" ]; then
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o pf -- python $OLDPWD/tools/prefill_bench.py mistral-7b fp8 2 128 > $OLDPWD/$OUT/prof.log 2>&1; cd $OLDPWD
  python tools/prof_summary.py $OUT/prof --tag pf_scratch > $OUT/prof_summary.md 2>> $OUT/prof.log; head -20 $OUT/prof_summary.md >> $OUT/summary.txt
fi
cat $OUT/summary.txt
