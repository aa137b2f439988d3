#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries committed under profiles/.

    python tools/prof_summary.py <kernel-trace dir> [<pmc dir>] --tag r01

* kernel trace (rocprofv3 --kernel-trace --stats): per-kernel calls / total / average / min duration.
  Handles both output flavours of this rocprofv3: CSV (*kernel_trace.csv) and the rocpd SQLite database.
* PMC pass (rocprofv3 --pmc FETCH_SIZE --kernel-trace): average FETCH_SIZE per launch per kernel, converted
  to bytes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE is in KiB and reports exactly
  half of a wide coalesced stream on gfx950: bytes = FETCH_SIZE * 1024 * 2).
* --bytes <json>: the library's own account of the ALGORITHMIC bytes each kernel was launched over (CALM_HIP_PROF_JSON, the
  analogue of the reference's PROF_TOKEN kernel argument, src/infer.cu:22,679) -- joined by kernel name into a GB/s column,
  the way tools/cudaprof.cu:85-100 turns its tokens into "BW (GB/s)"; with a PMC pass also the HBM bytes actually fetched.
Writes profiles/<tag>_kernel_stats.md and profiles/<tag>_pmc.json.
"""
import csv
import glob
import json
import os
import sqlite3
import statistics
import sys
from collections import defaultdict


def short(name):
    """`void (anonymous namespace)::k_pf_gemm_big<8, 0>(PfGemmArgs)` -> `k_pf_gemm_big<8, 0>`: namespaces (the anonymous one carries a
    parenthesis of its own -- round 4 cut the name there and filed 10 % of the trace under "") and the `void ` go, then the argument list"""
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("calm::", "")
    depth = 0
    for i, ch in enumerate(name):  # the argument list opens at the first "(" outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


def kernel_durations(d):
    durs = defaultdict(list)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    csvs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if csvs:
        for f in csvs:
            for row in csv.DictReader(open(f)):
                durs[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    elif dbs:
        for f in dbs:
            cur = sqlite3.connect(f).cursor()
            for name, start, end in cur.execute("select name, start, end from kernels"):
                durs[short(name)].append((end - start) / 1e3)
    return durs


def pmc_fetch(d):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == "FETCH_SIZE":
                out[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(f).cursor()
        try:
            rows = cur.execute("select name, counter_value from pmc_events where counter_name = 'FETCH_SIZE'")
            for name, v in rows:
                out[short(name)].append(float(v))
        except sqlite3.Error:
            pass
    return out


def base(name):
    return name.split("<")[0]


def main():
    flags = ("--tag", "--workload", "--bytes")
    args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] not in flags]
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r01"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    durs = kernel_durations(args[0])
    total = sum(sum(v) for v in durs.values())
    bytes_file = sys.argv[sys.argv.index("--bytes") + 1] if "--bytes" in sys.argv else None
    alg = json.load(open(bytes_file)) if bytes_file and os.path.exists(bytes_file) else {}  # (commands that never decode write no byte account)
    fetch = pmc_fetch(args[1]) if len(args) > 1 else {}
    fetch_by_base = defaultdict(list)
    for k, v in fetch.items():
        fetch_by_base[base(k)].extend(v)
    lines = [f"# rocprofv3 --kernel-trace summary ({tag}); durations in microseconds" + ("; algorithmic bytes from the library (CALM_HIP_PROF_JSON)" if alg else ""), "",
             "| kernel | calls | total us | % | avg us | median us | min us | algorithmic MB/launch | GB/s (algorithmic / avg) | HBM MB/launch (PMC, x2) | fetched / algorithmic |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        a = alg.get(base(k))
        mb = a["algorithmic_bytes"] / a["launches"] / 1e6 if a and a["launches"] else None
        avg = sum(v) / len(v)
        f = fetch_by_base.get(base(k))
        fmb = sum(f) / len(f) * 1024 * 2 / 1e6 if f else None
        lines.append(f"| `{k}` | {len(v)} | {sum(v):.0f} | {100*sum(v)/total:.1f} | {avg:.2f} | {statistics.median(v):.2f} | {min(v):.2f} | "
                     + (f"{mb:.2f}" if mb is not None else "") + " | " + (f"{mb / avg * 1e3:.0f}" if mb else "") + " | " + (f"{fmb:.2f}" if fmb is not None else "") + " | "
                     + (f"{fmb / mb:.3f}" if fmb is not None and mb else "") + " |")
    open(os.path.join(root, "profiles", f"{tag}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # the same table for programs (bench.py quotes the dominant kernel's rocprof average beside its own event timing), stamped
    # with the hash of the kernel sources it was taken on
    sys.path.insert(0, root)
    from calm_amd.build import csrc_sha as _sha

    js = {k: {"calls": len(v), "avg_us": sum(v) / len(v), "median_us": statistics.median(v), "min_us": min(v)} for k, v in durs.items() if k}
    js["_csrc_sha"] = _sha()
    if os.environ.get("CALM_HIP_GRAPH") == "0":  # (tools/hipprof.sh: traced with eager launches)
        js["_launch_mode"] = "eager launches (CALM_HIP_GRAPH=0): the same kernels, grids and arguments as the graph replays"
    js["_workload"] = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "mistral-7b fp8"
    json.dump(js, open(os.path.join(root, "profiles", f"{tag}_kernel_stats.json"), "w"), indent=1)
    if fetch:
        res = {}
        for k, v in fetch.items():
            res[k] = {"launches": len(v), "FETCH_SIZE_KiB_avg": sum(v) / len(v), "hbm_read_bytes_per_launch_corrected": sum(v) / len(v) * 1024 * 2}
        sys.path.insert(0, root)
        from calm_amd.build import csrc_sha

        res["_csrc_sha"] = csrc_sha()  # bench.py reports `traffic` from this file only while the kernel sources are these
        res["_workload"] = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "mistral-7b fp8"
        json.dump(res, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
        res = {k: v for k, v in res.items() if not k.startswith("_")}
        for k, r in sorted(res.items(), key=lambda kv: -kv[1]["hbm_read_bytes_per_launch_corrected"]):
            print(f"{k:40s} FETCH_SIZE avg {r['FETCH_SIZE_KiB_avg']:12.1f} KiB  -> {r['hbm_read_bytes_per_launch_corrected']/1e6:9.2f} MB/launch (x2 gfx950 correction)")


if __name__ == "__main__":
    main()
