#!/bin/bash
# workgroup shape / staging-first A/B: 256-thread workgroups (two per CU) vs one 512-thread workgroup per CU, with and without
# a barrier between the staging loads and the first weight tiles
OUT=gpurun_out/r03_m; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== quick parity of the 512-thread + staging-first variant" | tee $OUT/summary.txt
CALM_HIP_LIB=$PWD/calm_amd/libcalm_hip_wg512sf.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or greedy or moe" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
for rep in 1 2; do
for lib in libcalm_hip.so libcalm_hip_sf.so libcalm_hip_wg512.so libcalm_hip_wg512sf.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
done
echo "== timelines: product shape, then 512 + staging first" | tee -a $OUT/summary.txt
timeout 300 python tools/timeline.py 2>&1 | grep -E "^(kernel|qkv|attn_out|ffn_up|ffn_down|output|mistral)" >> $OUT/summary.txt
TL_TAG=wg512sf timeout 300 python tools/timeline.py 2>&1 | grep -E "^(kernel|qkv|attn_out|ffn_up|ffn_down|output|mistral)" >> $OUT/summary.txt
for cfg in "llama-3-8b gf4 8" "tinyllama-1.1b fp16 22"; do
for lib in libcalm_hip.so libcalm_hip_wg512sf.so; do
  echo "-- $lib $cfg" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py $cfg brief >> $OUT/summary.txt 2>&1
done
done
cat $OUT/summary.txt
