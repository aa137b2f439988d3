#!/bin/bash
# hipprof.sh -- the cudaprof analogue (reference tools/cudaprof.sh + tools/cudaprof.cu): run a command that drives libcalm_hip.so
# under rocprofv3 and print, per kernel, time share / average time / calls / BANDWIDTH over the algorithmic bytes the library
# says it launched it on (CALM_HIP_PROF_JSON plays the reference's PROF_TOKEN) / HBM bytes actually fetched (separate PMC pass).
#
#   tools/hipprof.sh [-t TAG] [-w "model dtype"] [--no-pmc] -- <command ...>        e.g.  -- python bench.py --no-cpu --steps 64
#
# Three runs of <command>: kernel trace (+ the byte account), FETCH_SIZE counters, nothing else; summaries land in
# profiles/TAG_kernel_stats.md and profiles/TAG_pmc.json (the latter stamped with the kernel sources' hash, which is what lets
# bench.py quote roofline.traffic from it).  The counter pass never shares a run with a trace domain other than the kernel trace.
# rocprofv3 7.2 has crashed on this library's hipGraph replays (SIGSEGV under hipGraphLaunch / "AQL packet is malformed": round 4 in
# decode_greedy_hip, round 5 in both passes over bench.py; not deterministic): trace with eager launches -- CALM_HIP_GRAPH=0 tools/hipprof.sh ...
# (the same kernels, grids and arguments) -- and note it in the summary JSON ("_launch_mode"; prof_summary.py does when the variable is set).
TAG=hipprof; WORKLOAD="mistral-7b fp8"; PMC=1
while [ $# -gt 0 ]; do
  case "$1" in
    -t) TAG=$2; shift 2;;
    -w) WORKLOAD=$2; shift 2;;
    --no-pmc) PMC=0; shift;;
    --) shift; break;;
    *) break;;
  esac
done
[ $# -gt 0 ] || { echo "usage: $0 [-t TAG] [-w 'model dtype'] [--no-pmc] -- command ..." >&2; exit 2; }
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CALM_HIP_PROF_JSON=$OUT/kernel_bytes.json timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
echo "kernel trace: exit $?"
if [ $PMC = 1 ]; then
  timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o fetch -- "$@" > $OUT/pmc.log 2>&1
  echo "FETCH_SIZE pass: exit $?"
  python $ROOT/tools/prof_summary.py $OUT/trace $OUT/pmc --tag $TAG --workload "$WORKLOAD" --bytes $OUT/kernel_bytes.json
else
  python $ROOT/tools/prof_summary.py $OUT/trace --tag $TAG --workload "$WORKLOAD" --bytes $OUT/kernel_bytes.json
fi
find $OUT/trace $OUT/pmc -type f -size +20M -delete 2>/dev/null
