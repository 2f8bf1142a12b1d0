"""Fused wi + GEGLU launch (gemm_nt8_kernel<bf16, false, 8, 1>) under the MRMT3_GEMM8_DBG store knock-outs: what its h / g stores cost
(1 no stores at all, 64 no h stores, 128 no g stores; results are wrong with any bit set)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the kernel diagnostics this tool switches on exist in the -DMRMT3_DIAG build only (make -C mr-mt3_amd/csrc diag)
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0"); lib.load()
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
M, K, dff = 65536, 512, 1024
a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(2 * dff, K, device=dev) * 0.05).bfloat16()
out = torch.empty(M, 2 * dff, device=dev, dtype=torch.bfloat16)
tiles = M // 256 * dff // 128 / 256
for p in (0.1, 0.0):
    for dbg, what in ((0, "as shipped"), (1, "no stores"), (64, "no h stores (g only: 134 MB)"), (128, "no g stores (h only: 268 MB)"), (192, "no h, no g stores (epilogue arithmetic only)")):
        os.environ["MRMT3_GEMM8_DBG"] = str(dbg)
        t = timeit(lambda: lib.gemm_nt_geglu(a, b, p=p, seed=1, stream_id=2))
        print(f"wi + GEGLU M={M} dff={dff} K={K} p={p} dbg={dbg:3d} {what:48s}: {t:7.1f} us, per tile {t/tiles:6.2f} us")
os.environ["MRMT3_GEMM8_DBG"] = "0"
t = timeit(lambda: lib.gemm_nt(a, b, out=out))
print(f"plain N = 2048 product (268 MB of C, whole lines)                                                  : {t:7.1f} us, per tile {t/tiles:6.2f} us")
