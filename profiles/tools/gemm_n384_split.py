"""N = 384 products (cq, d_o, d_co: 24 decoder + 8 encoder launches per step): one launch with a half-overlapping second column
tile (what the ping-pong kernel does today) against columns [0, 256) on the ping-pong kernel + columns [256, 384) on the round-1
256 x 128 tile kernel (two launches, A read twice).  Cold: 512 MiB written before every launch, as inside the step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0"); lib.load()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def cold(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for M, K in ((65536, 512), (16384, 512), (12288, 512)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(384, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, 384, device=dev, dtype=torch.bfloat16); out2 = torch.empty_like(out)
    one = lambda: lib.gemm_nt(a, w, out=out)
    def two():
        lib.gemm_nt(a, w[:256], out=out2[:, :256])
        lib.gemm_nt(a, w[256:], out=out2[:, 256:])
    t1, t2 = cold(one), cold(two)
    tl = cold(lambda: lib.gemm_nt(a, w[:256], out=out2[:, :256])); tr = cold(lambda: lib.gemm_nt(a, w[256:], out=out2[:, 256:]))
    one(); two(); torch.cuda.synchronize()
    print(f"M={M} N=384 K={K}: one launch {t1:6.1f} us | split 256 + 128: {t2:6.1f} us ({tl:.1f} + {tr:.1f}) | same bits: {torch.equal(out, out2)}")
