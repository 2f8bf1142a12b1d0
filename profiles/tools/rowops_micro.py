"""Row-wise kernels with and without dropout (is the mask generator or HBM the bound?).
   python profiles/tools/rowops_micro.py"""
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
M, d, dff = 65536, 512, 1024


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(M, d, device=dev)
y = torch.randn(M, d, device=dev).bfloat16()
w = torch.ones(d, device=dev)
h = torch.randn(M, 2 * dff, device=dev).bfloat16()
dg = torch.randn(M, dff, device=dev).bfloat16()
dxn = torch.randn(M, d, device=dev).bfloat16()
dres = torch.randn(M, d, device=dev).bfloat16()
dw = torch.zeros(d, device=dev)
x1, xn, rstd = lib.add_rmsnorm_fwd(x, y, w, 1e-6, torch.bfloat16)
for p in (0.0, 0.1):
    t1 = timeit(lambda: lib.add_rmsnorm_fwd(x, y, w, 1e-6, torch.bfloat16, p=p, seed=1, stream_y=3))
    t2 = timeit(lambda: lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, w, dw, p=p, seed=1, stream_y=3, dx1=dres))
    t3 = timeit(lambda: lib.geglu_fwd(h, p=p, seed=1, stream_id=5))
    t4 = timeit(lambda: lib.geglu_bwd(h, dg, p=p, seed=1, stream_id=5))
    print(f"p={p}: add_rmsnorm_fwd {t1:6.1f} us   add_rmsnorm_bwd {t2:6.1f} us   geglu_fwd {t3:6.1f} us   geglu_bwd {t4:6.1f} us")
