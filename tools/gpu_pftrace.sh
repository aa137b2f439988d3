#!/bin/bash
# rocprofv3 kernel trace of ONE prompt through prefill_hip (layer-reduced shape), summarised per kernel:
#   tools/gpu_pftrace.sh TAG MODEL DTYPE LAYERS TOKENS       -> gpurun_out/TAG/kernel_stats.md (+ printed)
TAG=$1; MODEL=${2:-mixtral-8x7b}; DTYPE=${3:-fp8}; L=${4:-4}; N=${5:-4096}
cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
PF_PROFILE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o pf -- python $R/tools/prefill_bench.py $MODEL $DTYPE $L $N > $R/$OUT/prof.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof --tag ${TAG}_scratch > /dev/null 2>> $OUT/prof.log
mv profiles/${TAG}_scratch_kernel_stats.md $OUT/kernel_stats.md 2>/dev/null
rm -f profiles/${TAG}_scratch_*
find $OUT -type f -size +20M -delete
tail -3 $OUT/prof.log; cut -c1-160 $OUT/kernel_stats.md | head -24
