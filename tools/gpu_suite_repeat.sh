#!/bin/bash
# The whole GPU suite, exactly as the driver runs it (python -m pytest tests/ -x -q -m gpu), N times in fresh processes on one box:
#   tools/gpu_suite_repeat.sh TAG N      -> gpurun_out/TAG/summary.txt: per run the exit code, the wall time and pytest's last lines
#                                           (a failure's whole report), and test_profiling's table-vs-stage ratios of every run
TAG=${1:-suite}; N=${2:-5}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export CALM_TEST_DIAG=$PWD/$OUT/diag.txt
echo "# $N runs of: python -m pytest tests/ -x -q -m gpu   (box: $(hostname), $(date -u +%FT%TZ), sources $(python -c 'from calm_amd.build import csrc_sha; print(csrc_sha())'))" > $OUT/summary.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 1200 python -m pytest tests/ -x -q -m gpu > $OUT/run_$i.log 2>&1
  rc=$?
  echo "== run $i: exit $rc after $(( $(date +%s) - t0 )) s" >> $OUT/summary.txt
  if [ $rc = 0 ]; then tail -1 $OUT/run_$i.log >> $OUT/summary.txt; else tail -60 $OUT/run_$i.log >> $OUT/summary.txt; fi
done
echo "== smoke" >> $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/summary.txt 2>&1
echo "== test_profiling: perf_hip's table vs perf_stage_hip, per run [GB/s table, GB/s here, ratio]" >> $OUT/summary.txt
cat $OUT/diag.txt >> $OUT/summary.txt 2>/dev/null
cat $OUT/summary.txt
