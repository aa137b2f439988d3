#!/bin/bash
# smallest GPU session: one tune sweep (args passed through), nothing else
TAG=${1:-mini}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 600 python tools/tune.py "$@" > $OUT/tune.txt 2>&1
cat $OUT/tune.txt
