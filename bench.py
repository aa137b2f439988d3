#!/usr/bin/env python3
"""bench.py -- the headline measurement: batch-1 greedy decode tok/s of a Mistral-7B-shaped fp8 .calm
model on MI355X through the drop-in C ABI (forward_hip per token + host argmax, exactly the loop of
reference src/run.c:167-256), with the roofline fraction of the dominant kernel and the reference CPU
path timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model mistral-7b] [--dtype fp8] [--pipeline [P]]

A "step" is one decode step (one token through every layer + classifier + sampling).  Data are
synthetic (seeded random weights of the real shapes; no checkpoints or network on the box).
N > 1: the path has one token in flight and does not shard for a model that fits one GPU, so ranks run
independent replicas (DESIGN.md section "multi-GPU"); value = tokens of all ranks / max-over-ranks time.
Started without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) it launches itself under
torch.distributed.run, one rank per GPU, so the printed n_gpus is N either way.
--gpus N --pipeline: BASELINE config 5 instead -- ONE process, the layers of DBRX-132B fp8 (or --model) split over P = N pipeline
stages inside the library (CALM_HIP_DEVICES), stage s on GPU s, one token in flight ("scaling": "capacity": the GPUs add
memory, not throughput).
A plain `--gpus N` run with N > 1 ALSO measures that: once the replicas are timed and have released their GPUs, rank 0 runs
`bench.py --gpus N --pipeline --no-cpu` as a child process and embeds its line as "pipeline": {...} (--no-pipeline-leg skips it),
so the one command the driver issues per N yields both the replica rate and config 5's number.
Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling there: 6290
HBM_NT_CEILING_GBPS = 6700.0  # this repo's own non-temporal streaming-read membench (profiles/r01_membench.txt)


def cpu_baseline(model, n_tokens, first_token, budget_s):
    """the reference CPU path (oracle/_ref = the untouched src/infer.c; our C restatement if that binary is absent) timed on
    this host on THE SAME model -- same depth, width, vocabulary and seeded weights as the GPU run -- decoding greedily from the
    same first token like src/run.c:167-256 does: up to n_tokens positions, cut short once budget_s seconds of CPU work are
    spent (the sample says how many were timed).  No extrapolation: value = timed positions / their wall time.
    Returns (record, reference tokens, reference logits of the first positions)."""
    from oracle import oracle  # test infrastructure: the reported CPU baseline and the parity checker, nothing else

    # thread count: the reference's own default is half the logical CPUs (src/infer.c:171-176); on the 256-thread GPU hosts
    # that many threads measured erratic (one OpenMP region per matmul), so the run is capped at 32 unless the caller says
    cores = int(os.environ.get("OMP_NUM_THREADS", max(min((os.cpu_count() or 2) // 2, 32), 1)))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    kind = "reference" if oracle.have_ref() else "port"
    be = oracle.RefBackend(model) if kind == "reference" else oracle.OracleBackend(model)
    be.forward(first_token, 0, 0)  # warm-up: pages in the weights (src/run.c:609-612)
    toks, keep, tok = [], [], first_token
    t0 = time.perf_counter()
    for pos in range(n_tokens):
        lg = be.forward(tok, pos, 0)
        if pos < 32:
            keep.append(lg.copy())
        tok = int(np.argmax(lg))
        toks.append(tok)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    be.close()
    n = len(toks)
    return {
        "value": round(n / dt, 3),
        "unit": "tok/s",
        "cores": cores,
        "kind": kind,
        "sample": f"the same {model.config.n_layers}-layer model and weights as the GPU run, greedy decode of its first {n} positions "
                  f"({dt:.1f} s of CPU, {dt / n * 1e3:.1f} ms per token)",
    }, toks, keep


def pipeline_leg(args, n_gpus):
    """BASELINE config 5 as a child process: `bench.py --gpus N --pipeline N --no-cpu` (DBRX-132B fp8 over N in-library stages) in a
    clean single-process environment; -> the sub-record embedded as "pipeline" in the replica line"""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "CALM_HIP_DEVICE", "OMP_NUM_THREADS")
           and not k.startswith("TORCHELASTIC")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n_gpus), "--pipeline", str(n_gpus), "--no-cpu", "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.dry_run:
        cmd.append("--dry-run")
    if args.layers:
        cmd += ["--layers", str(args.layers)]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=float(os.environ.get("CALM_BENCH_PIPELINE_TIMEOUT", "900")))
    except subprocess.TimeoutExpired:
        return {"error": "timeout", "command": " ".join(cmd[1:])}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"exit {r.returncode}", "stderr_tail": r.stderr[-600:], "command": " ".join(cmd[1:])}
    d = json.loads(lines[-1])
    keep = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "achieved_GBps", "hbm_frac_of_spec", "stage_devices", "handoff_us", "handoffs_timed", "dry_run",
            "steps_note", "load_seconds")
    rec = {k: d[k] for k in keep if k in d}
    rec["config"] = d.get("config")
    rec["command"] = "bench.py " + " ".join(cmd[2:])
    rec["wall_seconds"] = round(time.perf_counter() - t0, 1)
    return rec


# BASELINE.json's other GPU configurations (3, 4, 5; config 1 = TinyLlama is the reference's CPU plumbing case, measured here on the GPU
# as well): what the default N = 1 run measures after the headline workload, each in a child process of its own
OTHER_CONFIGS = [("llama-3-8b", "gf4", 0), ("tinyllama-1.1b", "fp16", 0), ("mixtral-8x7b", "fp8", 0), ("dbrx-132b", "fp8", 0), ("dbrx-132b", "fp8", 4)]


def other_configs_leg(args):
    """The reference publishes a table, not one number (README.md:88-107).  After the headline workload has released the GPU, every other
    BASELINE configuration runs as `bench.py --gpus 1 --model M --dtype D --steps 64 --no-cpu --no-extras` in a child process (weights
    synthesised on the device: 1-4 s per model), DBRX-132B also as 4 in-library pipeline stages on this one GPU; -> one record per
    workload.  The whole leg is bounded (CALM_BENCH_OTHER_BUDGET seconds, default 150): what did not fit is listed as dropped."""
    import subprocess

    budget = float(os.environ.get("CALM_BENCH_OTHER_BUDGET", "150"))
    t_leg = time.perf_counter()
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "CALM_HIP_DEVICES", "OMP_NUM_THREADS")
           and not k.startswith("TORCHELASTIC")}
    out = []
    for model, dtype, stages in OTHER_CONFIGS:
        name = f"{model} {dtype}" + (f", {stages} in-library pipeline stages on one GPU" if stages else "")
        left = budget - (time.perf_counter() - t_leg)
        if left < 20:
            out.append({"workload": name, "dropped": f"the leg's {budget:.0f} s budget was spent"})
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--model", model, "--dtype", dtype, "--steps", "64", "--warmup", "8", "--no-cpu", "--no-extras",
               "--no-other-configs"]
        if stages:
            cmd += ["--pipeline", str(stages)]
        if args.layers:
            cmd += ["--layers", str(args.layers)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left)
        except subprocess.TimeoutExpired:
            out.append({"workload": name, "dropped": f"timeout after {left:.0f} s (what was left of the leg's budget)"})
            continue
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            out.append({"workload": name, "error": f"exit {r.returncode}", "stderr_tail": r.stderr[-400:]})
            continue
        d = json.loads(lines[-1])
        rec = {"workload": name, "tok_s": d["value"], "steps": d["steps"], "ms_per_step": d["ms_per_step"], "achieved_GBps": d["achieved_GBps"],
               "hbm_frac_of_spec": d["hbm_frac_of_spec"], "bytes_per_step": d["bytes_per_step"],
               "dominant_kernel": {"kernel": d["roofline"]["kernel"], "us_per_launch": d["roofline"]["us_per_launch"], "achieved_GBps": d["roofline"]["achieved"],
                                   "frac": d["roofline"]["frac"]},
               "load_seconds": d["load_seconds"], "wall_seconds": round(time.perf_counter() - t0, 1)}
        if d.get("baseline_metric"):
            rec["tok_s_256"] = d["baseline_metric"]["tok_s"]
            rec["hbm_frac_of_spec_256"] = d["baseline_metric"]["hbm_frac_of_spec"]
        if stages:
            rec["stage_devices"], rec["handoff_us"] = d.get("stage_devices"), d.get("handoff_us")
        out.append(rec)
    return out


def steps_note(steps, with_256=False):
    """BASELINE.json's metric is a 256-token decode: a run with another step count averages over a different range of KV lengths"""
    if steps == 256:
        return None
    return (f"{steps} decode steps from position 0, not the 256 of BASELINE.json's metric: attention averages over KV lengths up to {steps} "
            f"instead of 256 (short runs read faster); "
            + ("the 256-step figure measured in this same run is under \"baseline_metric\"" if with_256 else "quote a 256-step run against the baseline"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default=None, help="a BASELINE shape (default mistral-7b; dbrx-132b with --pipeline)")
    ap.add_argument("--dtype", default="fp8")
    ap.add_argument("--layers", type=int, default=0, help="override depth (debug only; invalidates the metric)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="cap on the CPU baseline's timed region")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-device-greedy", action="store_true", help="skip the extra device-side greedy decode leg (rocprofv3 7.2 crashes in it)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pipeline", type=int, nargs="?", const=-1, default=0,
                    help="split the layers over P pipeline stages inside this one process (CALM_HIP_DEVICES=P: stage s on GPU s %% visible GPUs; "
                    "BASELINE config 5's partitioning) instead of running on one GPU; without a value P = --gpus")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: the launch / rendezvous / timing protocol only (gloo), for the CPU tests")
    ap.add_argument("--host-synth", action="store_true", help="without a CPU leg: synthesise the weights on the host and upload them (default: on the device)")
    ap.add_argument("--no-pipeline-leg", action="store_true", help="--gpus N > 1 without --pipeline: do not run BASELINE config 5 (the N-stage layer pipeline) afterwards")
    ap.add_argument("--no-other-configs", action="store_true", help="the default N = 1 run: do not measure the other BASELINE configurations afterwards (other_configs_leg)")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs beside the timed region: device-side greedy decode, prompt ingestion")
    args = ap.parse_args()
    if args.pipeline == -1:
        args.pipeline = args.gpus
    # the other BASELINE configurations ride behind the HEADLINE run only: no --model / --layers / --pipeline given, one GPU
    want_others = not args.no_other_configs and args.model is None and not args.layers and args.pipeline <= 1 and args.gpus == 1 and not args.dry_run
    if args.no_extras:
        args.no_device_greedy = True
    if args.model is None:
        args.model = "dbrx-132b" if args.pipeline > 1 else "mistral-7b"

    if args.gpus > 1 and args.pipeline <= 1 and "WORLD_SIZE" not in os.environ:
        # started bare: become `python -m torch.distributed.run --nproc-per-node N bench.py ...` (one rank per GPU), so that a
        # plain `python bench.py --gpus 8` measures 8 GPUs and says so
        import socket

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and args.pipeline <= 1 and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}", file=sys.stderr)
    if args.pipeline > 1:
        assert world == 1, "--pipeline is a single-process mode: launch without torchrun"
        os.environ["CALM_HIP_DEVICES"] = str(args.pipeline)  # read by init_hip
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("CALM_HIP_DEVICE", str(local_rank))
    if world > 1:
        # every rank synthesises its own copy of the weights with an OpenMP filler: share the host cores
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or world) // world)))

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_

        if args.dry_run:
            dist_.init_process_group("gloo")
        else:
            from calm_amd.host import require_torch_first

            require_torch_first("bench.py --gpus N")  # torch's bundled HIP runtime first, libcalm_hip.so's after
            torch.cuda.set_device(local_rank)
            dist_.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_

    if args.dry_run:
        # the protocol without the device: barrier, a stand-in for the timed region, max-over-ranks, one JSON line on rank 0
        from calm_amd.replicas import aggregate_throughput

        if dist is not None:
            dist.barrier()
        elapsed = 1e-3 * args.steps * (1 + 0.1 * rank)
        if dist is not None:
            dist.barrier()
        agg = aggregate_throughput(dist, args.steps, elapsed, device="cpu")
        if dist is not None:
            dist.barrier()
        if rank == 0:
            n_gpus = args.pipeline if args.pipeline > 1 else world
            line = {"metric": f"decode tok/s (batch=1, {args.steps} tok)", "value": round(agg["value"], 2), "unit": "tok/s", "n_gpus": n_gpus,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(agg["elapsed"] / args.steps * 1e3, 4),
                    "higher_is_better": True, "scaling": "capacity" if args.pipeline > 1 else "weak", "vs_baseline": None, "dtype": "f32",
                    "data": "none (dry run: no device work)", "dry_run": True, "steps_note": steps_note(args.steps),
                    "config": {"workload": f"{args.model} {args.dtype}", "parallelism": f"{args.pipeline}-stage layer pipeline" if args.pipeline > 1 else f"{world} replica(s)"}}
            if args.pipeline > 1:
                line["stage_devices"] = list(range(args.pipeline))
                line["handoff_us"] = None
            elif world > 1 and not args.no_pipeline_leg:
                line["pipeline"] = pipeline_leg(args, world)
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    from calm_amd import calmfile as cf
    from calm_amd.host import STAGES, HipBackend, HostModel, generate

    spec = cf.SPECS[args.model]
    n_layers = args.layers or spec.n_layers
    want_cpu = not args.no_cpu and world == 1  # the CPU baseline is an N = 1 measurement (rank 0 would otherwise hold the others up)
    cpu_skipped = None
    if want_cpu:
        # the CPU leg needs the whole model in host RAM (46.7 GB for Mixtral-8x7B fp8, 131.6 GB for DBRX-132B fp8)
        need = sum(a.nbytes for a in cf.stub_tensors(spec, args.dtype, n_layers).values()) + (6 << 30)
        try:
            import psutil

            have = int(psutil.virtual_memory().available)
        except ImportError:  # (not a dependency: fall back to the kernel's own account)
            have = int(next(ln.split()[1] for ln in open("/proc/meminfo") if ln.startswith("MemAvailable:"))) * 1024
        if have < need:
            cpu_skipped = f"host RAM: the CPU reference needs {need / 2**30:.0f} GiB for this model, {have / 2**30:.0f} GiB available"
            print(f"bench.py: no CPU leg ({cpu_skipped})", file=sys.stderr)
            want_cpu = False
    t0 = time.perf_counter()
    if want_cpu:
        # the CPU leg reads the same weights: hold the model on the host (every tensor its own array) and upload from there
        tensors, md = cf.synth_model_big(spec, args.dtype, args.seed, n_layers)
        model = HostModel(tensors, md)
        be = HipBackend(model)
    else:
        # no CPU leg: the same synthetic model is generated where it will live (tools/synth_fill_hip.hip; same bytes as the
        # host filler) -- what makes the 46.7 GB Mixtral-8x7B and 131.6 GB DBRX-132B shapes loadable in seconds
        model = HostModel(cf.stub_tensors(spec, args.dtype, n_layers), (cf.dataclasses.replace(spec, n_layers=n_layers)).metadata(args.dtype))
        if args.host_synth:
            be = HipBackend(model, stream=cf.synth_stream_big(spec, args.dtype, args.seed, n_layers))
        else:
            be = HipBackend(model, device_synth=(spec, args.dtype, args.seed, n_layers))
    load_s = time.perf_counter() - t0
    first_token = 17

    def barrier():
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

    # warm-up: W untimed decode steps (captures the hipGraphs, touches every weight once)
    generate(be, model, [first_token], args.warmup)
    barrier()
    t0 = time.perf_counter()
    toks, stats = generate(be, model, [first_token], args.steps)  # ends synchronised: forward_hip returned logits
    elapsed = time.perf_counter() - t0
    barrier()
    from calm_amd.replicas import aggregate_throughput

    agg = aggregate_throughput(dist, args.steps, elapsed)
    elapsed = agg["elapsed"]

    if rank != 0:
        be.close()  # (this rank's GPU is free again before rank 0 starts the pipeline leg)
        if dist is not None:
            barrier()
            dist.destroy_process_group()
        return

    n_params, n_bytes, n_bw = model.accounting()
    tok_s = agg["value"]
    step_bytes = stats["read_bytes"] / args.steps  # reference accounting: n_bandwidth + KV bytes (src/run.c:211-212)
    achieved = step_bytes * args.steps / elapsed / 1e9  # per GPU

    # BASELINE.json's own metric is a 256-token greedy decode (src/run.c:167-256, README.md:107).  A run with another --steps (the
    # driver's default is 20) averages attention over shorter KV lengths, so one 256-step decode of the same model is timed as well
    # (0.4 s for Mistral-7B) and reported beside `value`, which stays what the flags asked for.
    baseline_metric = None
    if args.steps != 256 and args.pipeline <= 1 and model.config.seq_len >= 256:
        generate(be, model, [first_token], 8)
        t0 = time.perf_counter()
        _, st256 = generate(be, model, [first_token], 256)
        dt256 = time.perf_counter() - t0
        gb256 = st256["read_bytes"] / dt256 / 1e9
        baseline_metric = {"metric": "decode tok/s (batch=1, 256 tok)", "steps": 256, "n_gpus": 1, "tok_s": round(256 / dt256, 2), "ms_per_step": round(dt256 / 256 * 1e3, 4),
                           "achieved_GBps": round(gb256, 1), "hbm_frac_of_spec": round(gb256 / HBM_PEAK_GBPS, 4),
                           "note": "one 256-step greedy decode from position 0 on rank 0's GPU, timed after the --steps region on the same resident model"}

    # roofline of the dominant kernel (FFN up: 2 x hidden x dim weight bytes per launch), HIP events on
    # the backend's stream, launches cycling over the layers so nothing is served from the Infinity Cache
    stage_report = {}
    if args.pipeline > 1:
        # a model split over stages is driven through forward_hip only: no per-stage timing; the roofline below is then the
        # whole step's algorithmic bytes over its time (every stage idles while the others work: one token in flight)
        dom = {"GBps": round(achieved, 1), "bytes": int(step_bytes), "us": round(elapsed / args.steps * 1e6, 2)}
    else:
        for i, name in enumerate(STAGES):
            us, b = be.stage_us(i, 8 if i != 5 else 1)
            stage_report[name] = {"us": round(us, 2), "GBps": round(b / us / 1e3, 1) if us > 0 else None, "bytes": int(b)}
        dom = stage_report["ffn_up"]
    # HBM bytes per launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE in its own
    # run, x1024 x2 per MI355X_MICROARCH.md; tools/prof_summary.py) -- only for the shape it was measured on
    traffic, traffic_source = None, None
    from calm_amd.build import csrc_sha

    for pmc_file in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        pmc = json.load(open(pmc_file))
        # a PMC pass describes the kernels it was taken on: used only when it carries the hash of the CURRENT kernel sources
        # and was taken on this shape; otherwise traffic is null rather than stale
        if pmc.get("_csrc_sha") != csrc_sha() or pmc.get("_workload") != f"{args.model} {args.dtype}":
            continue
        for k, v in pmc.items():
            if k.startswith("k_ffn_up<"):
                traffic = int(v["hbm_read_bytes_per_launch_corrected"])
                traffic_source = f"profiles/{os.path.basename(pmc_file)} (separate rocprofv3 --pmc FETCH_SIZE pass over the same kernel sources, x1024 x2)"
        break
    # the same kernel's average duration in the committed rocprofv3 --kernel-trace summary of this very source tree and shape
    # (profiles/r*_kernel_stats.json, tools/prof_summary.py), beside the live event-timed figure: the two must agree
    rocprof = None
    for ks_file in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.json")), reverse=True):
        ks = json.load(open(ks_file))
        if ks.get("_csrc_sha") != csrc_sha() or ks.get("_workload") != f"{args.model} {args.dtype}":
            continue
        for k, v in ks.items():
            if k.startswith("k_ffn_up<") and args.pipeline <= 1:
                gbps = dom["bytes"] / v["avg_us"] / 1e3
                rocprof = {"us_per_launch": round(v["avg_us"], 2), "achieved": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4), "calls": v["calls"],
                           # (the timed region above replays hipGraphs; rocprofv3 7.2 crashes on this library's long replays, so the trace is of
                           # eager launches of the same kernels, grids and arguments -- a kernel's own duration does not depend on that)
                           "launch_mode": (ks.get("_launch_mode") or "hipGraph replay").split(":")[0],
                           "source": f"profiles/{os.path.basename(ks_file)} (rocprofv3 --kernel-trace of the bench command, same kernel sources"
                                     + (f"; {ks['_launch_mode'].split(':')[0]}" if ks.get("_launch_mode") else "") + ")"}
        break
    roofline = {
        "bound": "hbm",
        "kernel": "k_ffn_up" if args.pipeline <= 1 else "whole decode step (all stages)",
        "achieved": dom["GBps"],
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(dom["GBps"] / HBM_PEAK_GBPS, 4),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "frac_of_measured_ceiling": round(dom["GBps"] / HBM_NT_CEILING_GBPS, 4),
        "bytes_per_launch": dom["bytes"],
        "us_per_launch": dom["us"],
        "rocprof": rocprof,
    }

    # device-side greedy decode of the same K tokens (no host round trip per token): extra, not `value`
    device_greedy = None
    if not args.no_device_greedy and args.pipeline <= 1:  # (decode_greedy_hip is a single-device entry point)
        be.decode_greedy(first_token, 0, min(args.steps, 8))  # captures its graphs
        t0 = time.perf_counter()
        dev_toks, _ = be.decode_greedy(first_token, 0, args.steps)
        dev_elapsed = time.perf_counter() - t0
        device_greedy = {"tok_s": round(args.steps / dev_elapsed, 2), "same_tokens": bool([int(t) for t in dev_toks] == toks)}

    # prompt ingestion (prefill_hip, the next scope row after the decode step): extra, not `value`.
    # MFMA-bound on the f16 matrix cores: every weight x activation product is two v_mfma_f32_32x32x16_f16 products (the fp32
    # activation as hi + lo binary16), so the peak in algorithmic FLOPs is half the 2.5 PF dense f16 peak (MI355X_MICROARCH.md)
    prefill = None
    if not args.no_device_greedy:
        n_pf = min(4095 if spec.n_experts else 2048, model.config.seq_len - 1)  # one chunk: 2048 tokens for a dense model, up to 4096 for a mixture of experts
        prompt = [int(t) for t in np.random.default_rng(args.seed).integers(0, spec.vocab_size, size=n_pf)]
        be.prefill(prompt[:64], 0)
        be.prefill(prompt, 0)
        t0 = time.perf_counter()
        be.prefill(prompt, 0)
        pf_elapsed = time.perf_counter() - t0
        q_dim, kv_dim = spec.n_heads * spec.head_dim, spec.n_kv_heads * spec.head_dim
        n_act = spec.n_experts_active if spec.n_experts else 1
        flop = 2.0 * n_pf * n_layers * (spec.dim * (2 * q_dim + 2 * kv_dim) + 3 * n_act * spec.dim * spec.hidden_dim)
        prefill = {"tokens": n_pf, "tok_s": round(n_pf / pf_elapsed, 1), "vs_serial_decode": round(n_pf / pf_elapsed / tok_s, 2),
                   "roofline": {"bound": "mfma", "achieved": round(flop / pf_elapsed / 1e12, 1), "peak": 1250.0, "unit": "TFLOP/s",
                                "frac": round(flop / pf_elapsed / 1e12 / 1250.0, 4),
                                "dtype": "f16 weights x (hi + lo) f16 activations, f32 accumulate; GEMM FLOPs only, attention time included"}}

    cpu = None
    parity = None
    if want_cpu:
        cpu, ref_toks, ref_logits = cpu_baseline(model, args.steps, first_token, args.cpu_seconds)
        # parity on the benchmark's own model at its full depth: the HIP logits teacher-forced along the reference's greedy
        # stream (first positions), and the two greedy streams side by side over every position the CPU leg decoded
        worst, tok = 0.0, first_token
        for pos, lc in enumerate(ref_logits):
            lg = be.forward(tok, pos, 0)
            worst = max(worst, float(np.abs(lg - lc).max() / np.abs(lc).max()))
            tok = ref_toks[pos]
        n_cmp = min(len(ref_toks), len(toks))
        first_diff = next((i for i in range(n_cmp) if toks[i] != ref_toks[i]), None)
        parity = {"sample": f"this run's {model.config.n_layers}-layer model: {len(ref_logits)} teacher-forced positions vs the CPU reference, "
                            f"greedy streams compared over {n_cmp} positions",
                  "max_rel_err": float(f"{worst:.3e}"), "tol": 1e-3, "greedy_identical": first_diff is None, "first_difference_at": first_diff}

    stage_devices, handoff_us, handoffs = None, None, None
    if args.pipeline > 1:
        # where the stages really sit, and what one hand-off of the residual stream costs: a few more steps with the library's
        # profiling on (an event pair around every stage-to-stage copy; perf_hip prints the same)
        stage_devices = [int(be.lib.calm_hip_query(b"stage_device", s_)) for s_ in range(be.lib.calm_hip_query(b"stages", 0))]
        be.lib.calm_hip_configure(b"prof", 1)
        tok = first_token
        for pos in range(8):
            tok = int(np.argmax(be.forward(tok, pos, 0)))
        be.lib.calm_hip_configure(b"prof", 0)
        handoffs = int(be.lib.calm_hip_query(b"handoffs", 0))
        handoff_us = round(be.lib.calm_hip_query(b"handoff_ns", 0) / 1e3, 2) if handoffs else None

    out = {
        "metric": f"decode tok/s (batch=1, {args.steps} tok)",
        "value": round(tok_s, 2),
        "unit": "tok/s",
        "n_gpus": min(args.pipeline, be.lib.calm_hip_device_count()) if args.pipeline > 1 else world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        # replicas: per-GPU work fixed as N grows; the layer pipeline: one model split over the GPUs -- they add capacity, not rate
        "scaling": "capacity" if args.pipeline > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.model} {args.dtype} .calm shape, batch 1, greedy {args.steps}-token decode via forward_hip + host argmax (run.c generate loop)",
            "weights": f"{args.dtype} ({cf.DBITS[args.dtype]} bit), fp32 activations/accumulate, fp16 KV cache",
            "n_layers": n_layers, "dim": spec.dim, "hidden_dim": spec.hidden_dim, "vocab": spec.vocab_size, "context": model.config.seq_len,
            "parallelism": (f"{args.pipeline}-stage layer pipeline in one process over {min(args.pipeline, be.lib.calm_hip_device_count())} GPU(s), one token in flight"
                            if args.pipeline > 1 else ("single GPU" if world == 1 else f"{world} independent replicas")),
            "device": be.lib.calm_hip_device_name().decode(),
        },
        "achieved_GBps": round(achieved, 1),
        "hbm_frac_of_spec": round(achieved / HBM_PEAK_GBPS, 4),
        "hbm_frac_of_measured_ceiling": round(achieved / HBM_NT_CEILING_GBPS, 4),
        "bytes_per_step": int(step_bytes),
        "roofline": roofline,
        "stages": stage_report,
        "device_greedy": device_greedy,
        "prefill": prefill,
        "cpu_baseline": cpu,
        "cpu_baseline_skipped": cpu_skipped,
        "parity": parity,
        "load_seconds": round(load_s, 1),
        "steps_note": steps_note(args.steps, baseline_metric is not None),
        "baseline_metric": baseline_metric,
    }
    if args.pipeline > 1:
        out["stage_devices"] = stage_devices
        out["handoff_us"] = handoff_us
        out["handoffs_timed"] = handoffs
    be.close()
    if dist is not None:
        barrier()  # every rank has released its GPU
        dist.destroy_process_group()
    if world > 1 and args.pipeline <= 1 and not args.no_pipeline_leg:
        # BASELINE config 5, which the replica run above is not: DBRX-132B fp8 over N in-library stages
        out["pipeline"] = pipeline_leg(args, world)
    if want_others and world == 1:
        out["other_configs"] = other_configs_leg(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
