"""Single-workgroup-per-CU latency of the attention kernels: B=4 gives 96 paired causal workgroups (< 256 CUs), so the
time is 18 key-tile iterations of ONE resident workgroup — the per-tile dependency-chain latency, not throughput."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
if os.environ.get("MRMT3_TOOL_LIB"):
    lib.LIB_PATH = os.environ["MRMT3_TOOL_LIB"]
dev = torch.device("cuda:0")
lib.load()
H, L = 6, 1024
for B in (4, 8, 16, 32, 64):
    qkv = torch.randn(B * L, 1152, device=dev).bfloat16()
    qkv[:, :384] *= 0.35
    q, k, v = qkv[:, :384], qkv[:, 384:768], qkv[:, 768:]
    for p in (0.0, 0.1):
        f = lambda: lib.attn_fwd(q, k, v, B, H, L, L, True, p=p, seed=1, stream_id=1)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        wgs = B * H * 4
        print(f"B={B:2d} p={p}: {wgs:4d} workgroups ({wgs/256:.2f}/CU)  fwd {t:7.1f} us  -> {t/18*1e3:6.0f} ns per tile-iteration if one round")
