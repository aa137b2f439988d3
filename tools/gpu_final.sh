#!/bin/bash
# Evidence session on the GPU box (outputs under gpurun_out/final_r03, summaries copied to profiles/ afterwards):
#  0. the whole GPU test suite as the driver runs it
#  2. tools/hipprof.sh over the bench command: rocprofv3 kernel trace + the library's algorithmic-byte account + FETCH_SIZE pass
#     (first: the bench line of section 1 then quotes roofline.traffic / roofline.rocprof from the files this pass stamps)
#  1. the default bench line (CPU reference timed on the same model, full-depth parity inside)
#  3. bench lines of the other BASELINE shapes at full size, each WITH its CPU leg and parity (host copy of the model: 46.7 GB for
#     Mixtral-8x7B, 131.6 GB for DBRX-132B); DBRX also as a 4-stage pipeline in one process (config 5's partitioning on one GPU)
#  4. the reference CLI on this backend, first 32 / last 32 positions of a 4096 context (CALM_POSO), full 32-layer Mistral shape
#  5. PMC counters of the gf4 kernels (tools/pmc_kernel.sh)
#  6. the long-context table (tools/longctx.py: Mistral geometry at 4k fp16 / 32k e5m2, DBRX geometry at 4k)
#  7. the real-architecture shape sweep with per-stage timings (tests/test_shape_sweep.py, CALM_SHAPE_SWEEP_OUT)
#  SECTIONS="1 2 3" selects (default: all)
TAG=${1:-r05}
SECTIONS=${SECTIONS:-"0 2 1 3 4 5 6 7"}
want() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
: > $OUT/summary.txt
if want 0; then
echo "== 0. pytest -m gpu" | tee -a $OUT/summary.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu.log 2>&1; echo "exit $?" >> $OUT/summary.txt
tail -6 $OUT/pytest_gpu.log >> $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/summary.txt 2>&1
fi
if want 2; then
echo "== 2. hipprof" | tee -a $OUT/summary.txt
# (eager launches: rocprofv3 7.2 dies on this library's hipGraph replays -- "AQL packet is malformed" / SIGSEGV, rounds 4-6; the same kernels, grids, arguments)
CALM_HIP_GRAPH=0 tools/hipprof.sh -t $TAG -w "mistral-7b fp8" -- python bench.py --no-cpu --no-device-greedy --no-other-configs --steps 64 --warmup 8 >> $OUT/summary.txt 2>&1
cp profiles/${TAG}_kernel_stats.md profiles/${TAG}_kernel_stats.json profiles/${TAG}_pmc.json $OUT/ 2>/dev/null
cp gpurun_out/$TAG/kernel_bytes.json $OUT/ 2>/dev/null
fi
if want 1; then
echo "== 1. bench" | tee -a $OUT/summary.txt
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" >> $OUT/summary.txt
tail -c 3800 $OUT/bench.json >> $OUT/summary.txt; tail -4 $OUT/bench.err >> $OUT/summary.txt
echo "== 1b. bench as the driver runs it (--steps 20): value for 20 steps + baseline_metric = the 256-step decode of the same run" | tee -a $OUT/summary.txt
( time timeout 600 python bench.py --steps 20 --warmup 3 ) > $OUT/bench_steps20.json 2> $OUT/bench20.err; echo "exit $?" >> $OUT/summary.txt
python - >> $OUT/summary.txt <<PY
import json
d = json.loads([l for l in open("$OUT/bench_steps20.json") if l.startswith("{")][-1])
print("value", d["value"], "tok/s at", d["steps"], "steps | baseline_metric", d["baseline_metric"], "| rocprof", (d["roofline"] or {}).get("rocprof"), "| traffic", d["roofline"]["traffic"])
PY
fi
if want 3; then
echo "== 3. other shapes at full size, CPU leg and parity included" | tee -a $OUT/summary.txt
rm -f $OUT/other_configs.jsonl
for cfg in "llama-3-8b gf4" "tinyllama-1.1b fp16" "mixtral-8x7b fp8" "dbrx-132b fp8"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --dtype $2 --steps 128 >> $OUT/other_configs.jsonl 2>> $OUT/other.err; echo "$cfg exit $?" >> $OUT/summary.txt
done
timeout 900 python bench.py --gpus 4 --pipeline --no-cpu --steps 128 >> $OUT/other_configs.jsonl 2>> $OUT/other.err; echo "dbrx pipeline 4 exit $?" >> $OUT/summary.txt
python - >> $OUT/summary.txt <<PY
import json
for l in open("$OUT/other_configs.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:34], "|", d["config"]["parallelism"][:34], "|", d["value"], "tok/s", d["achieved_GBps"], "GB/s", d["hbm_frac_of_spec"], "| cpu", (d.get("cpu_baseline") or {}).get("value"),
          "| parity", (d.get("parity") or {}).get("max_rel_err"), (d.get("parity") or {}).get("greedy_identical"), "| load", d["load_seconds"], "s")
PY
fi
if want 4; then
echo "== 4. reference CLI on the HIP backend: first / last 32 positions of a 4096 context" | tee -a $OUT/summary.txt
if [ -x oracle/_ref/run_hip ]; then
  python -m calm_amd.calmfile mistral-7b fp8 /tmp/mistral7b_fp8.calm >> $OUT/summary.txt 2>&1
  for poso in 0 4064; do
    echo "-- CALM_POSO=$poso" >> $OUT/summary.txt
    CALM_POSO=$poso timeout 300 oracle/_ref/run_hip /tmp/mistral7b_fp8.calm -i "abc" -t 0 -n 32 > $OUT/cli_poso_$poso.out 2> $OUT/cli_poso_$poso.err
    head -c 300 $OUT/cli_poso_$poso.out | head -2 >> $OUT/summary.txt; tail -2 $OUT/cli_poso_$poso.err >> $OUT/summary.txt
  done
  rm -f /tmp/mistral7b_fp8.calm
fi
fi
if want 5; then
echo "== 5. counters of the gf4 kernels (Llama-3-8B shape, full depth; three separate passes)" | tee -a $OUT/summary.txt
bash tools/pmc_kernel.sh ${TAG}_pmc_gf4 llama-3-8b gf4 > $OUT/pmc_gf4.log 2>&1
cp gpurun_out/${TAG}_pmc_gf4/summary.txt $OUT/pmc_gf4_tables.txt 2>/dev/null; tail -12 $OUT/pmc_gf4_tables.txt | cut -c1-400 >> $OUT/summary.txt
fi
if want 6; then
echo "== 6. long context (attention stage us, full-depth tok/s)" | tee -a $OUT/summary.txt
timeout 600 python tools/longctx.py 8 > $OUT/long_context.txt 2>&1
MODEL=dbrx-132b POS=4000 timeout 600 python tools/longctx.py 2 >> $OUT/long_context.txt 2>&1
cat $OUT/long_context.txt >> $OUT/summary.txt
fi
if want 7; then
echo "== 7. shape sweep" | tee -a $OUT/summary.txt
rm -f $OUT/shape_sweep.txt
CALM_SHAPE_SWEEP_OUT=$PWD/$OUT/shape_sweep.txt timeout 900 python -m pytest tests/test_shape_sweep.py -m gpu -q > $OUT/shape_sweep.log 2>&1; tail -2 $OUT/shape_sweep.log >> $OUT/summary.txt
cat $OUT/shape_sweep.txt >> $OUT/summary.txt
fi
cat $OUT/summary.txt
