mkdir -p gpurun_out/r03_e; export TMPDIR=/tmp; cd /tmp/ 2>/dev/null; cd $GRAFT_REPO_ROOT
timeout 300 python tools/longctx.py 4 > gpurun_out/r03_e/longctx.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r03_e/prof -o lc -- python tools/longctx.py 4 > gpurun_out/r03_e/prof.log 2>&1
python - <<'PY' > gpurun_out/r03_e/kernels.txt 2>&1
import glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r03_e/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("calm::", "")
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{n:48s} calls {len(v):6d} avg {sum(v)/len(v):8.2f} p10 {v[len(v)//10]:8.2f} p50 {v[len(v)//2]:8.2f} p90 {v[len(v)*9//10]:8.2f}")
PY
cat gpurun_out/r03_e/longctx.txt gpurun_out/r03_e/kernels.txt
find gpurun_out/r03_e/prof -type f -size +5M -delete
