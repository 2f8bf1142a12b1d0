import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0"); lib.load()
M, N, K = 65536, 1152, 512
a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
g = torch.randn(M, N, device=dev).bfloat16(); dw = torch.zeros(N, K, device=dev)
for _ in range(2):
    lib.gemm_nt(a, b, out=out)
    lib.gemm_tn(g, a, dw)
torch.cuda.synchronize()
print("done")
