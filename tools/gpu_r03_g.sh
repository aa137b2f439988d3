#!/bin/bash
OUT=gpurun_out/r03_g; mkdir -p $OUT; export TMPDIR=/tmp; cd "$(dirname "$0")/.."
echo "== attention parity" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "attn or attention or long_context or split" > $OUT/pytest.log 2>&1
echo "exit $?" >> $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
echo "== long context" | tee -a $OUT/summary.txt
timeout 300 python tools/longctx.py 4 >> $OUT/summary.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o lc -- python tools/longctx.py 4 > $OUT/prof.log 2>&1
python - >> $OUT/summary.txt 2>&1 <<'PY'
import sqlite3, collections, glob
for f in glob.glob("gpurun_out/r03_g/prof/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    d = collections.defaultdict(list)
    for name, start, end in cur.execute("select name, start, end from kernels"):
        d[name.split("(")[0].replace("void ", "").replace("calm::", "")].append((end - start) / 1e3)
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v.sort()
        print(f"{n:44s} calls {len(v):6d} avg {sum(v)/len(v):7.2f} p10 {v[len(v)//10]:7.2f} p50 {v[len(v)//2]:7.2f} p90 {v[len(v)*9//10]:7.2f}")
PY
find $OUT/prof -type f -size +2M -delete
cat $OUT/summary.txt
