"""gemm8 diagnostics: one shape under the MRMT3_GEMM8_DBG switches (1 no stores, 2 nt stores, 4 cache-hot loads,
8 no LDS fragment reads, 16 no loads)."""
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
os.environ["MRMT3_GEMM8"] = "1"
for M, N, K in ((65536, 2048, 512), (65536, 2048, 2048)):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tiles = M // 256 * N // 256 / 256
    for dbg, what in ((0, "as shipped"), (1, "no stores"), (2, "plain (not streaming) stores"), (32, "half of the stores"), (5, "no stores, cache-hot loads"), (9, "no stores, no fragment reads"),
                      (17, "no stores, loads off"), (25, "no stores, no reads, loads off (barriers + MFMA only)")):
        os.environ["MRMT3_GEMM8_DBG"] = str(dbg)
        t = timeit(lambda: lib.gemm_nt(a, b, out=out))
        print(f"M={M} N={N} K={K} dbg={dbg:2d} {what:52s}: {t:7.1f} us, per tile {t/tiles:6.2f} us, per K step {t/tiles/(K//64):5.2f} us")
os.environ["MRMT3_GEMM8_DBG"] = "0"
