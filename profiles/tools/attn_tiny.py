import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0"); lib.load()
B, H, L = 64, 6, 256
qkv = torch.randn(B * L, 1152, device=dev).bfloat16()
for _ in range(2):
    o, lse = lib.attn_fwd(qkv[:, :384], qkv[:, 384:768], qkv[:, 768:], B, H, L, L, False)
torch.cuda.synchronize()
print("done")
