#!/bin/bash
OUT=gpurun_out/r03_i; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee $OUT/summary.txt
tail -c 3500 $OUT/bench.json >> $OUT/summary.txt; tail -4 $OUT/bench.err >> $OUT/summary.txt
( time timeout 900 python bench.py --model mixtral-8x7b --steps 128 ) > $OUT/bench_mixtral.json 2> $OUT/bench_mixtral.err; echo "mixtral exit $?" | tee -a $OUT/summary.txt
tail -c 3000 $OUT/bench_mixtral.json >> $OUT/summary.txt; tail -4 $OUT/bench_mixtral.err >> $OUT/summary.txt
cat $OUT/summary.txt
