#!/bin/bash
# GPU-box session: does a HIP runtime knob move the per-launch fixed cost?  8-layer Mistral-7B fp8, graph replay tok/s.
TAG=${1:-envsweep}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
run() { # label, env assignments...
  local label="$1"; shift
  echo "== $label" >> $OUT/sweep.txt
  ( env "$@" timeout 120 python tools/tune.py mistral-7b fp8 8 brief 2>&1 | grep -E "graph=|GB/s" ) >> $OUT/sweep.txt
}
: > $OUT/sweep.txt
run "default" A=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "ROC_USE_FGS_KERNARG=0" ROC_USE_FGS_KERNARG=0
run "DEBUG_HIP_GRAPH_BATCH_SIZE=256" DEBUG_HIP_GRAPH_BATCH_SIZE=256
run "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0
run "GPU_FLUSH_ON_EXECUTION=1" GPU_FLUSH_ON_EXECUTION=1
run "default again" A=1
cat $OUT/sweep.txt
