#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, profiles.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== device" | tee $OUT/summary.txt
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; free -g | head -2) >> $OUT/summary.txt 2>&1

echo "== kernel + tiny-model parity tests" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_hip_parity.py::test_mistral7b_full_depth_properties \
    -k "not full_width" > $OUT/pytest_small.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -15 $OUT/pytest_small.log >> $OUT/summary.txt

echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/smoke.log >> $OUT/summary.txt

echo "== full-width / full-depth parity tests" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q -k "full_width or full_depth" > $OUT/pytest_big.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -15 $OUT/pytest_big.log >> $OUT/summary.txt

echo "== bench" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/summary.txt
tail -c 3000 $OUT/bench.json >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt

echo "== overlap experiment" | tee -a $OUT/summary.txt
timeout 300 python tools/exp_overlap.py > $OUT/exp_overlap.txt 2>&1; cat $OUT/exp_overlap.txt >> $OUT/summary.txt

echo "== tune" | tee -a $OUT/summary.txt
timeout 600 python tools/tune.py > $OUT/tune.txt 2>&1; cat $OUT/tune.txt >> $OUT/summary.txt

echo "== rocprof kernel trace of the bench command" | tee -a $OUT/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --no-cpu --no-device-greedy --steps 64 --warmup 8 > $OUT/prof_bench.log 2>&1
echo "exit $?" >> $OUT/summary.txt
echo "== rocprof PMC pass (FETCH_SIZE), its own run" | tee -a $OUT/summary.txt
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o fetch -- python bench.py --no-cpu --no-device-greedy --steps 8 --warmup 2 > $OUT/prof_pmc.log 2>&1
echo "exit $?" >> $OUT/summary.txt
python tools/prof_summary.py $OUT/prof $OUT/pmc --tag $TAG >> $OUT/summary.txt 2>&1
mkdir -p $OUT/profiles && cp profiles/${TAG}_* $OUT/profiles/ 2>/dev/null

echo "== prompt ingestion: rates, kernel trace, MFMA counters" | tee -a $OUT/summary.txt
for cfg in "mistral-7b fp8 4 1024" "mistral-7b fp8 4 4000" "mistral-7b fp8 4 64" "llama-3-8b gf4 4 512" "tinyllama-1.1b fp16 8 512" "mixtral-8x7b fp8 2 512" "dbrx-132b fp8 1 512"; do
  timeout 300 python tools/prefill_bench.py $cfg >> $OUT/prefill.txt 2>&1
done
cat $OUT/prefill.txt >> $OUT/summary.txt
R=$PWD
(cd /tmp && PF_PROFILE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/pfprof -o pf -- python $R/tools/prefill_bench.py mistral-7b fp8 2 256 > $R/$OUT/pfprof.log 2>&1)
python tools/prof_summary.py $OUT/pfprof --tag ${TAG}_prefill > $OUT/prefill_kernel_stats.md 2>> $OUT/pfprof.log
cp profiles/${TAG}_prefill_kernel_stats.md $OUT/profiles/ 2>/dev/null
grep -E "kernel \||k_pf|true>" $OUT/prefill_kernel_stats.md >> $OUT/summary.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" "FETCH_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && PF_PROFILE=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pfpmc/$n -o pmc -- python $R/tools/prefill_bench.py mistral-7b fp8 2 256 > $R/$OUT/pfpmc_$n.log 2>&1)
done
python tools/pmc_table.py $OUT/pfpmc | grep -E "^kernel|k_pf|true>" > $OUT/profiles/${TAG}_prefill_pmc.txt 2>&1
cat $OUT/profiles/${TAG}_prefill_pmc.txt >> $OUT/summary.txt
find $OUT/pfprof $OUT/pfpmc -type f -size +30M -delete 2>/dev/null
# the raw traces are large: keep only what the summary needs
find $OUT/prof $OUT/pmc -type f -size +30M -delete 2>/dev/null
cat $OUT/summary.txt
