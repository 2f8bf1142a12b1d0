"""Headroom check for the NT / TN GEMM kernels: the same shapes through torch.matmul (hipBLASLt / rocBLAS on this image),
cold (512 MiB written before every timed launch, as profiles/tools/gemm_ab.py cold).  Not a product path — a yardstick.
Usage: python profiles/tools/blaslt_compare.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = torch.device("cuda:0")
lib.load()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
Md, Me = 65536, 16384
NT = [("qkv", Md, 1152, 512), ("o/co", Md, 512, 384), ("cq", Md, 384, 512), ("wi", Md, 2048, 512), ("wo", Md, 512, 1024),
      ("d_qkv", Md, 512, 1152), ("d_wi", Md, 512, 2048), ("d_wo", Md, 1024, 512), ("e_qkv", Me, 1152, 512),
      ("e_wi", Me, 2048, 512), ("e_dwi", Me, 512, 2048)]
TN = [("w_qkv", Md, 1152, 512), ("w_wi", Md, 2048, 512), ("w_wo", Md, 512, 1024), ("w_o", Md, 512, 384)]


def cold(fn):
    fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        flush.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ev)
    return t[len(t) // 2] * 1e-3


for name, M, N, K in NT:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wt = w.t()
    t_ours = cold(lambda: lib.gemm_nt(a, w, out=out))
    t_lib = cold(lambda: torch.matmul(a, wt, out=out))
    f = 2.0 * M * N * K
    print(f"NT {name:6s} M={M:5d} N={N:4d} K={K:4d}: ours {t_ours*1e6:7.1f} us {f/t_ours/1e12:6.0f} TF | torch.matmul {t_lib*1e6:7.1f} us {f/t_lib/1e12:6.0f} TF | ours/torch x{t_lib/t_ours:.2f}")
for name, M, N1, N2 in TN:
    dy = torch.randn(M, N1, device=dev).bfloat16()
    x = torch.randn(M, N2, device=dev).bfloat16()
    out = torch.empty(N1, N2, device=dev, dtype=torch.float32)
    outb = torch.empty(N1, N2, device=dev, dtype=torch.bfloat16)
    dyt = dy.t()
    t_ours = cold(lambda: lib.gemm_tn(dy, x, out=out))
    t_lib = cold(lambda: torch.matmul(dyt, x, out=outb))
    f = 2.0 * M * N1 * N2
    print(f"TN {name:6s} M={M:5d} {N1:4d}x{N2:4d}: ours (single launch) {t_ours*1e6:7.1f} us {f/t_ours/1e12:6.0f} TF | torch.matmul {t_lib*1e6:7.1f} us {f/t_lib/1e12:6.0f} TF | x{t_lib/t_ours:.2f}")
