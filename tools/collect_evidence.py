#!/usr/bin/env python3
"""Copy the summaries of the round's evidence session (tools/gpu_final.sh TAG -> gpurun_out/final_<tag>/) into the tracked profiles/
directory, and build the gf4 counter table before / after from the previous round's table (profiles/r<NN-1>_pmc_gf4_tables.txt) and
the session's counter pass.     python tools/collect_evidence.py r04"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prev = f"r{int(tag[1:]) - 1:02d}"
F = os.path.join(ROOT, "gpurun_out", f"final_{tag}")
P = os.path.join(ROOT, "profiles")
for f in (f"{tag}_kernel_stats.md", f"{tag}_kernel_stats.json", f"{tag}_pmc.json"):
    if os.path.exists(os.path.join(F, f)):
        shutil.copy(os.path.join(F, f), os.path.join(P, f))
if os.path.exists(os.path.join(F, "kernel_bytes.json")):
    shutil.copy(os.path.join(F, "kernel_bytes.json"), os.path.join(P, f"{tag}_kernel_bytes.json"))
for src, dst in (("bench.json", f"{tag}_bench.json"), ("bench_steps20.json", f"{tag}_bench_steps20.json")):
    if os.path.exists(os.path.join(F, src)):
        line = [l for l in open(os.path.join(F, src)) if l.startswith("{")][-1]
        open(os.path.join(P, dst), "w").write(line)
if os.path.exists(os.path.join(F, "other_configs.jsonl")):
    shutil.copy(os.path.join(F, "other_configs.jsonl"), os.path.join(P, f"{tag}_other_configs_full_size.jsonl"))
if os.path.exists(os.path.join(F, "cli_poso_0.out")):
  with open(os.path.join(P, f"{tag}_reference_cli_on_hip.txt"), "w") as o:
    o.write(f"Round {int(tag[1:])}: the reference's own CLI (src/run.c, built as oracle/_ref/run_hip against libcalm_hip.so) on the HIP backend, Mistral-7B fp8 shape at full depth\n"
            "(tools/gpu_final.sh section 4; CALM_POSO shifts the positions: the first and the last 32 positions of a 4096-token context)\n")
    for p in (0, 4064):
        o.write(f"-- CALM_POSO={p}\n")
        o.write(open(os.path.join(F, f"cli_poso_{p}.out")).read()[:400] + "\n")
        o.write("".join(open(os.path.join(F, f"cli_poso_{p}.err")).readlines()[-3:]))
if os.path.exists(os.path.join(F, "pytest_gpu.log")):  # (round 5: the suite's record is profiles/r05_gpu_tests.txt, from tools/gpu_suite_repeat.sh)
  with open(os.path.join(P, f"{tag}_gpu_tests.txt"), "w") as o:
    o.write(f"Round {int(tag[1:])}: pytest -m gpu on the MI355X box (tools/gpu_final.sh section 0)\n")
    o.write("".join(open(os.path.join(F, "pytest_gpu.log")).readlines()[-8:]))
    o.write("".join(l for l in open(os.path.join(F, "summary.txt")) if "smoke ok" in l))


for src, dst in (("long_context.txt", f"{tag}_long_context_table.txt"), ("shape_sweep.txt", f"{tag}_shape_sweep.txt")):
    if os.path.exists(os.path.join(F, src)):
        # the tracked file = the session's raw table + whatever reading of it was written underneath (kept across re-collections)
        old = open(os.path.join(P, dst)).read() if os.path.exists(os.path.join(P, dst)) else ""
        notes = old[old.index("\n== reading"):] if "\n== reading" in old else ""
        open(os.path.join(P, dst), "w").write(open(os.path.join(F, src)).read().rstrip("\n") + "\n" + notes)


def table(lines):
    d = {}
    h = [l for l in lines if l.startswith("kernel |")][0].split(" | ")
    for l in lines:
        if l.startswith("k_") and " | " in l:
            f = l.split(" | ")
            d[f[0].split("<")[0]] = (f[0], dict(zip(h[1:], [float(x) if x != "-" else None for x in f[1:]])))
    return d


pm = os.path.join(F, "pmc_gf4_tables.txt")
if os.path.exists(pm):
    before = open(os.path.join(P, f"{prev}_pmc_gf4_tables.txt")).read().splitlines()
    after = open(pm).read().splitlines()
    B, A = table(before), table(after)
    out = [f"Round {int(tag[1:])}: counters of the gf4 kernels BEFORE (profiles/{prev}_pmc_gf4_tables.txt, the previous round's kernels) and AFTER (this round's),",
           "Llama-3-8B gf4 shape at full depth: tools/pmc_kernel.sh (bench.py under rocprofv3 --pmc, three separate passes).  Per launch, averaged over the",
           "launches of the run; SQ_* cycle counters in quad-cycles summed over the waves (MI355X_MICROARCH.md); FETCH_SIZE in its raw unit",
           "(x 64 B x 2 on gfx950 = bytes).", ""]
    cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
            "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM_RD", "FETCH_SIZE"]
    for k in ["k_qkv", "k_attn_out", "k_ffn_up", "k_ffn_down", "k_output"]:
        out.append(f"{k}:   before {B[k][0]}   after {A[k][0]}")
        out.append("   counter                  before        after    after/before")
        for c in cols:
            b, a = B[k][1].get(c), A[k][1].get(c)
            out.append(f"   {c:22s} {b:12.0f} {a:12.0f}   {a / b if b else float('nan'):6.2f}")
        b, a = B[k][1], A[k][1]
        out.append(f"   share of wave cycles: issuing {b['SQ_ACTIVE_INST_ANY'] / b['SQ_WAVE_CYCLES']:.2f} -> {a['SQ_ACTIVE_INST_ANY'] / a['SQ_WAVE_CYCLES']:.2f}, issue stalls "
                   f"{b['SQ_WAIT_INST_ANY'] / b['SQ_WAVE_CYCLES']:.2f} -> {a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']:.2f}, waiting (memory / barrier) "
                   f"{b['SQ_WAIT_ANY'] / b['SQ_WAVE_CYCLES']:.2f} -> {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.2f};  busy cycles {b['SQ_BUSY_CYCLES']:.0f} -> {a['SQ_BUSY_CYCLES']:.0f}")
        out.append("")
    out.append("raw table (after):")
    out += [l for l in after if l.startswith("kernel |") or l.startswith("k_")]
    open(os.path.join(P, f"{tag}_pmc_gf4_tables.txt"), "w").write("\n".join(out) + "\n")
print("collected into", P)
