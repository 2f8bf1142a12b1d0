"""A/B of the fused wi projection + gated GELU (mrmt3_gemm_nt_geglu) against gemm_nt + geglu_fwd on the step's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mr-mt3_amd"))
from mrmt3 import lib
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows in (32768, 65536):
    x = torch.randn(rows, 512, device=dev).bfloat16()
    wi = (torch.randn(2048, 512, device=dev) * 0.05).bfloat16()
    for p in (0.0, 0.1):
        os.environ["MRMT3_GEGLU_FUSED"] = "0"
        t0 = timeit(lambda: lib.gemm_nt_geglu(x, wi, p=p, seed=1, stream_id=3))
        tg = timeit(lambda: lib.gemm_nt(x, wi))
        os.environ["MRMT3_GEGLU_FUSED"] = "1"
        t1 = timeit(lambda: lib.gemm_nt_geglu(x, wi, p=p, seed=1, stream_id=3))
        os.environ["MRMT3_GEMM8_DBG"] = "1"
        t2 = timeit(lambda: lib.gemm_nt_geglu(x, wi, p=p, seed=1, stream_id=3))
        os.environ["MRMT3_GEMM8_DBG"] = "0"
        fl = 2.0 * rows * 2048 * 512
        print("rows %6d p %.1f: two kernels %7.1f us (gemm alone %7.1f us, %5.0f TF) | fused %7.1f us (no stores %7.1f) | saved %6.1f us"
              % (rows, p, t0, tg, fl / tg / 1e6, t1, t2, t0 - t1), flush=True)
