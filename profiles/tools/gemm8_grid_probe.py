import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the kernel diagnostics this tool switches on exist in the -DMRMT3_DIAG build only (make -C mr-mt3_amd/csrc diag)
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0"); lib.load()
def timeit(fn, reps=10):
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
M, N, K = 65536, 2048, 512
a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for grid in (256, 128, 64, 32):
    os.environ["MRMT3_GEMM8_GRID"] = str(grid)
    r = []
    for dbg in ("0", "1"):
        os.environ["MRMT3_GEMM8_DBG"] = dbg
        r.append(timeit(lambda: lib.gemm_nt(a, b, out=out)))
    tiles = 2048 / grid
    print(f"grid {grid:3d}: {r[0]:8.1f} us with stores, {r[1]:8.1f} without -> per tile {r[0]/tiles:6.2f} / {r[1]/tiles:6.2f} us (stores +{(r[0]-r[1])/tiles:5.2f})")
