#!/bin/bash
# counter passes over one shape of an experiment binary: tools/experiments/pmc_exp.sh TAG <command ...>
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$n -o pmc -- "$@" > $OUT/$n.log 2>&1
  echo "== $set : exit $?"
done
python tools/pmc_table.py $OUT | python -c "
import sys
rows=[l.rstrip('\n').split(' | ') for l in sys.stdin]
h=rows[0]
for r in rows[1:]:
    print(r[0])
    for n,v in zip(h[1:],r[1:]): print(f'   {n:28s} {v}')
"
find $OUT -type f -size +20M -delete
