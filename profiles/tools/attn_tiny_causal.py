"""One launch each of the decoder self-attention kernels of the training step (batch 64, causal, dropout 0.1) for
single-kernel PMC passes:  rocprofv3 --pmc <counters> --kernel-trace -- python3 profiles/tools/attn_tiny_causal.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0"); lib.load()
B, H, L = 64, 6, 1024
qkv = torch.randn(B * L, 1152, device=dev).bfloat16()
qkv[:, :384] *= 0.35
d_o = torch.randn(B * L, 384, device=dev).bfloat16()
q, k, v = qkv[:, :384], qkv[:, 384:768], qkv[:, 768:]
o, lse = lib.attn_fwd(q, k, v, B, H, L, L, True, p=0.1, seed=1, stream_id=1)
dq = torch.empty(B * L, 384, device=dev, dtype=torch.bfloat16); dk = torch.empty_like(dq); dv = torch.empty_like(dq)
lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, L, True, p=0.1, seed=1, stream_id=1)
torch.cuda.synchronize()
print("done")
