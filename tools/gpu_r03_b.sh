#!/bin/bash
# Round 3, second GPU session: engine gate second cut, preload A/B with the k_qkv regression fixed, torch / HIP runtime coexistence,
# the fixed tests, Mixtral + DBRX full-depth parity.
TAG=${1:-r03_b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== engine gate (exp_engine2, interleaved slots)" | tee $OUT/summary.txt
timeout 300 python tools/exp_engine2.py noedge > $OUT/engine2.txt 2>&1; echo "exit $?" >> $OUT/summary.txt
cat $OUT/engine2.txt >> $OUT/summary.txt
echo "== preload A/B: stage timings (tune.py brief)" | tee -a $OUT/summary.txt
for lib in libcalm_hip.so libcalm_hip_nopreload.so libcalm_hip.so libcalm_hip_nopreload.so; do
  echo "-- $lib" >> $OUT/summary.txt
  CALM_HIP_LIB=$PWD/calm_amd/$lib timeout 300 python tools/tune.py mistral-7b fp8 8 brief >> $OUT/summary.txt 2>&1
done
echo "== torch and libcalm_hip.so in one process" | tee -a $OUT/summary.txt
timeout 300 python - >> $OUT/summary.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
print("torch first: cuda available", torch.cuda.is_available(), "hip", torch.version.hip, "devices", torch.cuda.device_count())
from calm_amd.host import load_lib
lib = load_lib(); lib.init_hip(); print("then libcalm_hip:", lib.calm_hip_device_count(), lib.calm_hip_device_name())
x = torch.ones(4, device="cuda") * 2; print("torch tensor on gpu:", x.sum().item())
PY
timeout 300 python - >> $OUT/summary.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from calm_amd.host import load_lib
lib = load_lib(); lib.init_hip(); print("libcalm_hip first:", lib.calm_hip_device_count())
import torch
print("then torch: cuda available", torch.cuda.is_available(), "devices", torch.cuda.device_count())
PY
echo "== parity tests touched this round" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "not full_depth and not full_width and not long_context and not device_synth" > $OUT/pytest_small.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -6 $OUT/pytest_small.log >> $OUT/summary.txt
echo "== Mixtral-8x7B / DBRX-132B fp8 full depth" | tee -a $OUT/summary.txt
( time timeout 1800 python -m pytest tests/test_full_depth_moe.py -q -x -s ) > $OUT/pytest_moe.log 2>&1
echo "exit $?" >> $OUT/summary.txt; grep -E "fp8|passed|failed|Error|real|DBRX" $OUT/pytest_moe.log | tail -12 >> $OUT/summary.txt
cat $OUT/summary.txt
