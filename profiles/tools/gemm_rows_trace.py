"""Per-workgroup phase timeline of a fused projection + row launch (mrmt3_gemm_rows_trace): when workgroups start, how long
their K loop / hand-off / row phase take, and how the two workgroups of a CU overlap.
    python3 profiles/tools/gemm_rows_trace.py [rows = 65536] [K = 384]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the kernel diagnostics this tool switches on exist in the -DMRMT3_DIAG build only (make -C mr-mt3_amd/csrc diag)
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import numpy as np
import torch
from mrmt3 import lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dev = torch.device("cuda:0")
L = lib.load()
import ctypes
L.mrmt3_gemm_rows_trace.restype, L.mrmt3_gemm_rows_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]   # (diagnostics build only: not in the header)
PADW, PADA = int(os.environ.get("PADW", "0")), int(os.environ.get("PADA", "0"))
a = torch.randn(M, K + PADA, device=dev).bfloat16()[:, :K]
w = (torch.randn(512, K + PADW, device=dev) * K ** -0.5).bfloat16()[:, :K]
x0 = torch.randn(M, 512, device=dev)
wn = torch.ones(512, device=dev)
x1 = torch.empty_like(x0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
BM = 128 if (os.environ.get('MRMT3_ROWS_BM', '0') == '128' or (os.environ.get('MRMT3_ROWS_BM', '0') != '64' and -(-M // 128) >= 256)) else 64
grid = -(-M // BM)
trace = torch.zeros(grid, 8, dtype=torch.int64, device=dev)
for it in range(3):
    flush.zero_()
    if it == 2:
        L.mrmt3_gemm_rows_trace(trace.data_ptr())
    lib.gemm_nt_addnorm(a, w, x0, wn, 1e-6, p=0.1, seed=1, stream_y=3, x1=x1)
    torch.cuda.synchronize()
L.mrmt3_gemm_rows_trace(None)
t = trace.cpu().numpy().astype(np.float64) * 0.01          # us
t0 = t[:, 0].min()
t -= t0
start, kbeg, kend, img, end = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4]
print("rows %d K %d: %d workgroups; launch span %.1f us" % (M, K, grid, end.max()))
if len(sys.argv) > 3:            # compact
    first = start < 1.0
    kl = (kend - kbeg)[first]
    print("   padw %d pada %d dbg %s: first-wave K loop min %.1f median %.1f max %.1f us; row phase median %.1f" % (
        PADW, PADA, os.environ.get("MRMT3_ROWS_DBG", "0"), kl.min(), np.median(kl), kl.max(), np.median((end - img)[first])))
    sys.exit(0)
def q(x):
    return "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
print("start time        ", q(start))
print("K loop            ", q(kend - kbeg))
print("hand-off          ", q(img - kend))
print("row phase         ", q(end - img))
print("workgroup lifetime", q(end - start))
first = start < np.percentile(start, 45)
print("first wave of workgroups (%d): K loop %s" % (first.sum(), q((kend - kbeg)[first])))
print("                             row phase %s" % q((end - img)[first]))
print("later workgroups (%d): start %s" % ((~first).sum(), q(start[~first])))
print("                       K loop %s" % q((kend - kbeg)[~first]))
print("                       row phase %s" % q((end - img)[~first]))
# how many workgroups are in which phase over time
for tt in np.linspace(0, end.max(), 17)[1:-1]:
    ink = ((kbeg <= tt) & (tt < kend)).sum()
    inr = ((img <= tt) & (tt < end)).sum()
    print("t = %6.1f us: %4d in the K loop, %4d in the row phase, %4d not started, %4d done" % (tt, ink, inr, (start > tt).sum(), (end <= tt).sum()))
