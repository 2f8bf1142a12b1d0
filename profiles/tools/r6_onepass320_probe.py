"""Round 6, VERDICT r5 item 4: the one-pass cross-attention backward at 320 keys (MR-MT3's own model) against the two-pass kernels, one
site of the 64-segment step: B = 64, H = 6, Lq = 1024, Lk = 320, dropout 0.1; and the 256-key site before / after the generalisation.
Kill criterion of the review: above 180 us (two-pass: ~242)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mr-mt3_amd")):
    sys.path.insert(0, p)
import torch

from mrmt3 import lib

dev = torch.device("cuda", 0)
B, H, Lq = 64, 6, 1024


def site(Lk, p, onepass, reps=30):
    g = torch.Generator(device="cpu").manual_seed(5)
    q = (torch.randn(B * Lq, H * 64, generator=g) * 0.35).to(dev).bfloat16()
    k = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    v = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    d_o = torch.randn(B * Lq, H * 64, generator=g).to(dev).bfloat16()
    o, lse, o_lo = lib.attn_fwd(q, k, v, B, H, Lq, Lk, False, p=p, seed=31, stream_id=4, want_lo=True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.set_knob("MRMT3_ATTN_ONEPASS", 1 if onepass else 0)
    scratch = torch.empty(128 << 20, dtype=torch.float32, device=dev)
    ts = []
    for i in range(reps + 3):
        scratch.fill_(float(i))                       # 512 MiB written: operands out of the caches, as inside the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, False, p=p, seed=31, stream_id=4, o_lo=o_lo)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    lib.reset_knobs()
    return ts[len(ts) // 2], dq, dk, dv


for Lk in (256, 320):
    for p in (0.1, 0.0):
        t1, a1, b1, c1 = site(Lk, p, True)
        t0, a0, b0, c0 = site(Lk, p, False)
        rel = max(float((x.float() - y.float()).norm() / y.float().norm()) for x, y in ((a1, a0), (b1, b0), (c1, c0)))
        print("Lk = %3d  dropout %.1f   one-pass %7.1f us   two-pass %7.1f us   ratio %.2f   max rel diff of dQ / dK / dV %.2e"
              % (Lk, p, t1, t0, t0 / t1, rel), flush=True)
