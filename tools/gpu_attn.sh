#!/bin/bash
TAG=${1:-attn}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -m gpu -q -k "attention or split or golden or kv" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
LONGCTX=1 timeout 600 python tools/tune.py mistral-7b fp8 8 brief > $OUT/tune.txt 2>&1; cat $OUT/tune.txt
