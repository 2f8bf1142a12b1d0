"""One grouped weight-gradient launch over every TN site of a training step (batch 64): the workload of the PMC passes
in profiles/tools/pmc_tn_group.sh.  Prints the plan and the event-timed duration; under rocprofv3 --pmc the counters of
gemm_tn8_group_kernel / tn8_group_reduce_kernel give the HBM traffic of the whole family in one row each."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

dev = torch.device("cuda:0")
Md, Me = 65536, 16384
dec = [("w_wo", Md, 512, 1024), ("w_wi", Md, 2048, 512), ("w_co", Md, 512, 384), ("w_cq", Md, 384, 512),
       ("w_ckv", Me, 768, 512), ("w_o", Md, 512, 384), ("w_qkv", Md, 1152, 512)]
enc = [("e_wo", Me, 512, 1024), ("e_wi", Me, 2048, 512), ("e_o", Me, 512, 384), ("e_qkv", Me, 1152, 512)]
sites = [("w_lm", Md, 1536, 512)] + dec * 8 + enc * 8 + [("e_proj", Me, 512, 512)]
grp = lib.TnGroup()
keep, flops, alg = [], 0.0, 0.0
for name, M, N1, N2 in sites:
    a = torch.randn(M, N1, device=dev).mul_(0.1).bfloat16()
    b = torch.randn(M, N2, device=dev).mul_(0.1).bfloat16()
    out = torch.zeros(N1, N2, device=dev)
    keep.append((a, b, out))
    flops += 2.0 * M * N1 * N2
    alg += (a.numel() + b.numel()) * 2 + out.numel() * 8        # operands once, dW read + written (accumulate)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
times = []
for r in range(reps):
    for a, b, out in keep:
        grp.add(a, b, out, accumulate=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    grp.flush()
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
info = grp.last_info
res = dict(sites=len(sites), flops=flops, algorithmic_bytes=alg, n_items=info.n_items, n_rtiles=info.n_rtiles,
           rounds=info.rounds, slab_bytes=int(info.slab_bytes), ms=times)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pmc_tn_group_plan.json"), "w"))
print(json.dumps(res))
print("grouped TN: %.3f ms best of %d -> %.0f TFLOP/s; partial tiles %.0f MB written + read" % (
    min(times), reps, flops / min(times) / 1e9, info.slab_bytes / 1e6))
