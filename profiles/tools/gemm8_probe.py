"""Where does a tile's time go in gemm8?  Times a few shapes with the C stores switched off (MRMT3_GEMM8_DBG=1) and
over a K sweep at fixed M, N (slope = one K step, intercept = per-tile + per-launch overhead)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the kernel diagnostics this tool switches on exist in the -DMRMT3_DIAG build only (make -C mr-mt3_amd/csrc diag)
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0")
lib.load()

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best

os.environ["MRMT3_GEMM8"] = "1"
for M, N in ((65536, 512), (65536, 2048), (65536, 1024)):
    for K in (128, 256, 512, 1024, 2048):
        a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = []
        for dbg in ("0", "1"):
            os.environ["MRMT3_GEMM8_DBG"] = dbg
            res.append(timeit(lambda: lib.gemm_nt(a, b, out=out)))
        tiles = (M // 256) * (N // 256) / 256
        print(f"M={M} N={N} K={K:4d}: {res[0]:7.1f} us with stores, {res[1]:7.1f} us without  ({tiles:.0f} tiles/WG, nk={K//64}) "
              f"-> per tile {res[0]/tiles:5.2f} / {res[1]/tiles:5.2f} us")
