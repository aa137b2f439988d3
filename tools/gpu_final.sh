#!/bin/bash
# GPU-box session that refreshes the round's evidence: the default bench line, the rocprofv3 kernel trace of the bench
# command and its FETCH_SIZE pass (separate run), summarised into profiles/${TAG}_*.
TAG=${1:-r01}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== bench" | tee $OUT/summary.txt
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/summary.txt
tail -c 3200 $OUT/bench.json >> $OUT/summary.txt; tail -6 $OUT/bench.err >> $OUT/summary.txt
echo "== rocprof kernel trace of the bench command" | tee -a $OUT/summary.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --no-cpu --no-device-greedy --steps 64 --warmup 8 > $OUT/prof_bench.log 2>&1
echo "exit $?" >> $OUT/summary.txt
echo "== rocprof PMC pass (FETCH_SIZE), its own run" | tee -a $OUT/summary.txt
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o fetch -- python bench.py --no-cpu --no-device-greedy --steps 8 --warmup 2 > $OUT/prof_pmc.log 2>&1
echo "exit $?" >> $OUT/summary.txt
python tools/prof_summary.py $OUT/prof $OUT/pmc --tag $TAG >> $OUT/summary.txt 2>&1
mkdir -p $OUT/profiles && cp profiles/${TAG}_kernel_stats.md profiles/${TAG}_pmc.json $OUT/profiles/ 2>/dev/null
find $OUT/prof $OUT/pmc -type f -size +20M -delete 2>/dev/null
cat $OUT/summary.txt
