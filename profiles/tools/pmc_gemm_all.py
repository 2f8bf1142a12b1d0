"""One launch of every NT / TN GEMM shape of the training step (batch 64, dtypes as the engine uses them) for the PMC
passes of profiles/tools/pmc_traffic.sh; writes the launch plan (order, launches per step, algorithmic bytes / flops)
next to the counters so that pmc_traffic_parse.py can pair dispatches with shapes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

dev = torch.device("cuda:0")
L = lib.load()
Md, Me = 65536, 16384
NT = [("qkv", Md, 1152, 512, "bf16", 8), ("o/co", Md, 512, 384, "bf16", 16), ("cq", Md, 384, 512, "bf16", 8),
      ("ckv", Me, 768, 512, "bf16", 8), ("wi", Md, 2048, 512, "bf16", 8), ("wo", Md, 512, 1024, "bf16", 8),
      ("lm_head", Md, 1536, 512, "f32", 1), ("d_lm", Md, 512, 1536, "f32", 1),
      ("d_qkv", Md, 512, 1152, "bf16", 8), ("d_wi", Md, 512, 2048, "bf16", 8), ("d_wo", Md, 1024, 512, "bf16", 8),
      ("d_o/co", Md, 384, 512, "bf16", 16), ("d_cq", Md, 512, 384, "bf16", 8),
      ("e_qkv", Me, 1152, 512, "bf16", 8), ("e_o", Me, 512, 384, "bf16", 8), ("e_wi", Me, 2048, 512, "bf16", 8),
      ("e_wo", Me, 512, 1024, "bf16", 8), ("e_dqkv", Me, 512, 1152, "bf16", 8), ("e_dwi", Me, 512, 2048, "bf16", 8),
      ("e_dwo", Me, 1024, 512, "bf16", 8), ("e_do", Me, 384, 512, "bf16", 8)]
TN = [("w_qkv", Md, 1152, 512, 8), ("w_o/co", Md, 512, 384, 16), ("w_cq", Md, 384, 512, 8), ("w_wi", Md, 2048, 512, 8),
      ("w_wo", Md, 512, 1024, 8), ("w_lm", Md, 1536, 512, 1), ("w_ckv", Me, 768, 512, 8), ("e_qkv", Me, 1152, 512, 8),
      ("e_o", Me, 512, 384, 8), ("e_wi", Me, 2048, 512, 8), ("e_wo", Me, 512, 1024, 8)]
plan = []
for name, M, N, K, od, per in NT:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if od == "bf16" else torch.float32)
    torch.cuda.synchronize()
    lib.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    plan.append(dict(kind="nt", name=name, per_step=per, flops=2.0 * M * N * K,
                     bytes=a.numel() * 2 + b.numel() * 2 + out.numel() * out.element_size()))
batch = lib.TnBatch()
keep = []          # the deferred reduce writes into every `out` at the end: they must stay allocated
for name, M, N1, N2, per in TN:
    a = torch.randn(M, N1, device=dev).bfloat16()
    b = torch.randn(M, N2, device=dev).bfloat16()
    out = torch.zeros(N1, N2, device=dev)
    keep.append(out)
    torch.cuda.synchronize()
    lib.gemm_tn(a, b, out, accumulate=True, defer=batch)      # the MFMA kernel alone (slabs)
    torch.cuda.synchronize()
    splits = L.mrmt3_gemm_tn_splits(M, N1, N2)
    plan.append(dict(kind="tn", name=name, per_step=per, flops=2.0 * M * N1 * N2, splits=splits,
                     bytes=a.numel() * 2 + b.numel() * 2 + out.numel() * 4, slab_bytes=splits * N1 * N2 * 4))
batch.flush()
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(plan, open(os.path.join(ROOT, "gpurun_out", "pmc_plan.json"), "w"))
print("launched", len(plan), "GEMMs")
