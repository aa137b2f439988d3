#!/bin/bash
# kernel trace + PMC passes of the prompt-ingestion path (layer-reduced Mistral-7B fp8, 128 tokens)
TAG=${1:-pfprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
CMD="python $R/tools/prefill_bench.py mistral-7b fp8 2 128"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o pf -- $CMD > $R/$OUT/prof.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof --tag pf_scratch > $OUT/kernel_stats.md 2>> $OUT/prof.log
grep -E "k_pf|k_attn<16, 16, true|kernel \|" $OUT/kernel_stats.md | tee $OUT/summary.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  cd /tmp; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/$n -o pmc -- $CMD > $R/$OUT/$n.log 2>&1; echo "== $set : exit $?" >> $R/$OUT/summary.txt; cd $R
done
python tools/pmc_table.py $OUT | grep -E "^kernel|k_pf|k_attn<16, 16, true" >> $OUT/summary.txt 2>&1
find $OUT -type f -size +20M -delete
cat $OUT/summary.txt
