#!/bin/bash
OUT=gpurun_out/r03_ad; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== long context: lane arithmetic / matrix cores with V transposed through LDS / the same with the transposition faked (timing only)" | tee $OUT/summary.txt
echo "-- product (k_attn_gqa)" >> $OUT/summary.txt
timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
echo "-- k_attn_mfma" >> $OUT/summary.txt
CALM_HIP_ATTN_MFMA=1 timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
echo "-- k_attn_mfma, V transposition faked" >> $OUT/summary.txt
CALM_HIP_LIB=$PWD/calm_amd/libcalm_hip_fakevt.so CALM_HIP_ATTN_MFMA=1 timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
