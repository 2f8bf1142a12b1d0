"""One launch of every NT / TN GEMM shape of the training step (batch 64), for PMC passes:
   rocprofv3 --pmc FETCH_SIZE -d out -- python3 profiles/tools/gemm_tiny_all.py   (then again with WRITE_SIZE)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

dev = torch.device("cuda:0")
lib.load()
Md, Me = 65536, 16384
NT = [("qkv", Md, 1152, 512, "bf16"), ("o", Md, 512, 384, "bf16"), ("cq", Md, 384, 512, "bf16"),
      ("ckv", Me, 768, 512, "bf16"), ("wi", Md, 2048, 512, "bf16"), ("wo", Md, 512, 1024, "bf16"),
      ("lm_head", Md, 1536, 512, "f32"), ("d_qkv", Md, 512, 1152, "bf16"), ("d_wi", Md, 512, 2048, "bf16"),
      ("d_wo", Md, 1024, 512, "bf16"), ("d_o", Md, 384, 512, "bf16")]
TN = [("w_qkv", Md, 1152, 512), ("w_o", Md, 512, 384), ("w_wi", Md, 2048, 512), ("w_wo", Md, 512, 1024),
      ("w_lm", Md, 1536, 512), ("w_ckv", Me, 768, 512)]
for name, M, N, K, od in NT:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if od == "bf16" else torch.float32)
    torch.cuda.synchronize()
    lib.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    print("NT", name, M, N, K, od, "algorithmic MB", (a.numel() * 2 + b.numel() * 2 + out.numel() * out.element_size()) / 1e6)
for name, M, N1, N2 in TN:
    a = torch.randn(M, N1, device=dev).bfloat16()
    b = torch.randn(M, N2, device=dev).bfloat16()
    out = torch.zeros(N1, N2, device=dev)
    torch.cuda.synchronize()
    lib.gemm_tn(a, b, out, accumulate=True)
    torch.cuda.synchronize()
    print("TN", name, M, N1, N2, "algorithmic MB", (a.numel() * 2 + b.numel() * 2 + out.numel() * 4) / 1e6)
