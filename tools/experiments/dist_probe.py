import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel, generate
from calm_amd.replicas import aggregate_throughput
spec = cf.SPECS["mistral-7b"]; L = 4
model = HostModel(cf.stub_tensors(spec, "fp8", L), cf.dataclasses.replace(spec, n_layers=L).metadata("fp8"))
be = HipBackend(model, device_synth=(spec, "fp8", 1, L))
generate(be, model, [17], 8)
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter(); toks, st = generate(be, model, [17], 64); el = time.perf_counter() - t0
torch.cuda.synchronize(); dist.barrier()
print("agg", aggregate_throughput(dist, 64, el))
be.close(); dist.barrier(); dist.destroy_process_group(); print("dist probe ok")
