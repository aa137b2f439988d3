#!/bin/bash
OUT=gpurun_out/r03_am; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== chunks of 3 .. 8 prompt tokens through one weight stream (k_pf_skinny)" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "prefill or perplexity" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -8 $OUT/pytest.log >> $OUT/summary.txt
timeout 300 python tools/smallchunk_bench.py mistral-7b fp8 8 >> $OUT/summary.txt 2>&1
timeout 300 python tools/smallchunk_bench.py llama-3-8b gf4 8 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
