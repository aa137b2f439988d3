#!/bin/bash
# PMC pass over one layer-reduced shape: why is a kernel not at the HBM rate?  usage: tools/pmc_kernel.sh <tag> <model> <dtype> <layers>
TAG=${1:-pmc}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$n -o pmc -- python tools/tune.py "$@" brief > $OUT/$n.log 2>&1
  echo "== $set : exit $?" | tee -a $OUT/summary.txt
done
python tools/pmc_table.py $OUT >> $OUT/summary.txt 2>&1
find $OUT -type f -size +20M -delete
cat $OUT/summary.txt
