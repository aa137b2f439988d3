#!/bin/bash
# PMC passes over one BASELINE shape: why is a kernel not at the HBM rate?  usage: tools/pmc_kernel.sh <tag> <model> <dtype>
# Three separate counter passes (never together with a trace domain other than the kernel trace) over a short bench run; each under
# its own short timeout, and the session stops at the first pass that fails (a counter pass that hangs costs its whole timeout).
TAG=${1:-pmc}; MODEL=${2:-mistral-7b}; DTYPE=${3:-fp8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
: > $OUT/summary.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$n -o pmc -- python bench.py --model $MODEL --dtype $DTYPE --no-cpu --no-device-greedy --steps 24 --warmup 4 > $OUT/$n.log 2>&1
  rc=$?
  echo "== $set : exit $rc" | tee -a $OUT/summary.txt
  [ $rc = 0 ] || break
done
python tools/pmc_table.py $OUT >> $OUT/summary.txt 2>&1
find $OUT -type f -size +20M -delete
cat $OUT/summary.txt
