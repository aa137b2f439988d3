#!/bin/bash
# PMC passes over tools/experiments/exp_pfgemm (the prompt GEMMs stand-alone): where the wide kernel's cycles go
OUT=gpurun_out/${1:-pfgpmc}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  cd /tmp; timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/$n -o pmc -- $R/tools/experiments/exp_pfgemm 1024 2 > $R/$OUT/$n.log 2>&1; echo "== $set : exit $?"; cd $R
done
python tools/pmc_table.py $OUT | grep -E "^kernel|k_pf_gemm" | tee $OUT/table.txt
find $OUT -type f -size +5M -delete
