#!/bin/bash
# Round 3, first GPU session: host facts, kernel-argument preload A/B (stage timings + wave timelines), the parity tests that
# changed this round, Mixtral-8x7B full-depth parity, and -- last, it is the one that can misbehave -- the engine gate experiment.
TAG=${1:-r03_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
{
  echo "== host"; nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|Thread" ; cat /sys/kernel/mm/transparent_hugepage/enabled
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "total" | head -2
} > $OUT/host.txt 2>&1
cat $OUT/host.txt
echo "== preload A/B: stage timings (tune.py brief)" | tee $OUT/summary.txt
for lib in libcalm_hip.so libcalm_hip_nopreload.so libcalm_hip.so libcalm_hip_nopreload.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
echo "== timelines" | tee -a $OUT/summary.txt
timeout 300 python tools/timeline.py >> $OUT/summary.txt 2>&1
timeout 300 python tools/timeline.py --no-preload >> $OUT/summary.txt 2>&1
echo "== parity tests touched this round" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "not full_depth and not full_width and not long_context and not device_synth" > $OUT/pytest_small.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -6 $OUT/pytest_small.log >> $OUT/summary.txt
echo "== Mixtral-8x7B fp8 full depth" | tee -a $OUT/summary.txt
( time timeout 1200 python -m pytest tests/test_full_depth_moe.py -q -x -s -k mixtral ) > $OUT/pytest_mixtral.log 2>&1
echo "exit $?" >> $OUT/summary.txt; grep -E "mixtral-8x7b fp8:|passed|failed|Error|real" $OUT/pytest_mixtral.log | tail -8 >> $OUT/summary.txt
echo "== engine gate (exp_engine2)" | tee -a $OUT/summary.txt
timeout 300 python tools/exp_engine2.py noedge > $OUT/engine2.txt 2>&1; echo "exit $?" >> $OUT/summary.txt
cat $OUT/engine2.txt >> $OUT/summary.txt
cat $OUT/summary.txt
