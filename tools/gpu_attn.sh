#!/bin/bash
# GPU-box session for attention-kernel work: the attention / split / golden parity tests, stage sweep, long-context table.
TAG=${1:-attn}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests -m gpu -q -x -k "attention or split or golden or kv or pipeline" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
if [ "${TUNE:-1}" = "1" ]; then timeout 200 python tools/tune.py mistral-7b fp8 8 brief > $OUT/tune.txt 2>&1; cat $OUT/tune.txt; fi
if [ "${LONG:-1}" = "1" ]; then timeout 300 python tools/longctx.py 8 > $OUT/longctx.txt 2>&1; cat $OUT/longctx.txt; fi
