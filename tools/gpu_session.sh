#!/bin/bash
# One GPU session: tools/gpu_session.sh TAG 'command' ['command' ...]   (run through gpurun from the repo root)
# Every command runs under its own timeout (STEP_TIMEOUT seconds, default 600) with stdout + stderr appended to
# gpurun_out/TAG/summary.txt, which is printed at the end (gpurun shows the tail) and merged back by gpurun.  Replaces the
# per-session scripts of rounds 1-3 (tools/gpu_r03_*.sh): the session IS its command line, recorded in the summary's header.
TAG=$1; shift
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
{ echo "# session $TAG"; for c in "$@"; do echo "#   $c"; done; } > $OUT/summary.txt
for c in "$@"; do
  echo "== $c" >> $OUT/summary.txt
  t0=$(date +%s)
  timeout ${STEP_TIMEOUT:-600} bash -c "$c" >> $OUT/summary.txt 2>&1
  echo "[exit $? after $(( $(date +%s) - t0 )) s]" >> $OUT/summary.txt
done
cat $OUT/summary.txt
