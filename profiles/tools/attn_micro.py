"""Micro-benchmark of the attention shapes of the training step, dropout off / on.
Usage: python profiles/tools/attn_micro.py [reps = 10] [segments = 64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
if os.environ.get("MRMT3_TOOL_LIB"):      # tuning tool only: A/B a variant build of the library
    lib.LIB_PATH = os.environ["MRMT3_TOOL_LIB"]

dev = torch.device("cuda:0")
lib.load()
H = 6
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, Lq, Lk, causal in (("dec-self", 1024, 1024, True), ("dec-cross", 1024, 256, False), ("enc-self", 256, 256, False)):
    qkv = torch.randn(B * Lq, 1152, device=dev).bfloat16()
    qkv[:, :384] *= 0.35
    kv = torch.randn(B * Lk, 768, device=dev).bfloat16()
    q = qkv[:, :384]
    k, v = (qkv[:, 384:768], qkv[:, 768:]) if Lq == Lk else (kv[:, :384], kv[:, 384:])
    d_o = torch.randn(B * Lq, 384, device=dev).bfloat16()
    for p in (0.0, 0.1):
        o, lse = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=1, stream_id=1)
        dq = torch.empty(B * Lq, 384, device=dev, dtype=torch.bfloat16)
        dk = torch.empty(B * Lk, 384, device=dev, dtype=torch.bfloat16)
        dv = torch.empty_like(dk)
        tf = timeit(lambda: lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=1, stream_id=1))
        tb = timeit(lambda: lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, causal, p=p, seed=1, stream_id=1))
        fl = 4.0 * B * H * Lq * Lk * 64
        print(f"{name:9s} p={p}: fwd {tf*1e6:8.1f} us ({fl/tf/1e12:6.1f} TF alg)   bwd {tb*1e6:8.1f} us ({2*fl/tb/1e12:6.1f} TF alg)")
