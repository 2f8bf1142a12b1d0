"""VERDICT r4 item 5, step 1 (the probe, with its kill criterion): can the weight-gradient products run UNDER the attention
backward instead of after it?  Two streams, eager launches, the 64-segment step's shapes:

  stream A   one decoder layer's attention backward — the causal self-attention site (attn_bwd_dq + attn_bwd_dkdv, 1024 x
             1024) and the cross-attention site (the one-pass kernel, 1024 x 256) — N_LAYERS times
  stream B   the same layers' weight gradients (d_wo, d_wi, d_co, d_cq, d_o, d_qkv: 65536-row TN products) with (a) the
             round-1 128x128 tile kernel (146 VGPRs x 4 waves, 64 KiB LDS: the one the soak showed co-resident with other
             kernels; knob MRMT3_TN8=0) and (b) the production ping-pong TN kernel (236 VGPRs x 8 waves, 128 KiB LDS)

and t(A || B) against t(A) + t(B).  KILL if t(A || B) > 0.85 x the sum: then the two do not share CUs in any useful way
and a small-footprint grouped TN variant has nothing to win.
    python3 profiles/tools/wgrad_under_attention.py [layers = 4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

N_LAYERS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
lib.load()
B, H, L, Le, d, inner, dff = 64, 6, 1024, 256, 512, 384, 1024
rows = B * L
torch.manual_seed(0)
step = torch.zeros(1, dtype=torch.int32, device=dev)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).bfloat16()


# ---- stream A's work: saved tensors of one self-attention site and one cross-attention site
qkv = rnd(rows, 3 * inner, scale=0.35)
o_s, lse_s, lo_s = lib.attn_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, H, L, L, True, p=0.1, seed=1,
                                step=step, stream_id=3, want_lo=True)
do_s = rnd(rows, inner)
dqkv = torch.empty_like(qkv)
q_c = rnd(rows, inner, scale=0.35)
kv_c = rnd(B * Le, 2 * inner, scale=0.35)
o_c, lse_c, lo_c = lib.attn_fwd(q_c, kv_c[:, :inner], kv_c[:, inner:], B, H, L, Le, False, p=0.1, seed=1, step=step,
                                stream_id=4, want_lo=True)
do_c = rnd(rows, inner)
dq_c, dkv_c = torch.empty_like(q_c), torch.empty_like(kv_c)


def attention_backward():
    for _ in range(N_LAYERS):
        lib.attn_bwd(q_c, kv_c[:, :inner], kv_c[:, inner:], o_c, do_c, lse_c, dq_c, dkv_c[:, :inner], dkv_c[:, inner:], B, H, L,
                     Le, False, p=0.1, seed=1, step=step, stream_id=4, o_lo=lo_c)
        lib.attn_bwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], o_s, do_s, lse_s, dqkv[:, :inner],
                     dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:], B, H, L, L, True, p=0.1, seed=1, step=step, stream_id=3,
                     o_lo=lo_s)


# ---- stream B's work: one decoder layer's weight gradients (a^T b, 65536 rows): (N1, N2)
WG = [(d, dff), (2 * dff, d), (d, inner), (inner, d), (d, inner), (3 * inner, d)]      # d_wo, d_wi, d_co, d_cq, d_o, d_qkv
ops = [(rnd(rows, n1, scale=0.1), rnd(rows, n2, scale=0.1), torch.zeros(n1, n2, device=dev)) for n1, n2 in WG]
flops_b = sum(2.0 * rows * n1 * n2 for n1, n2 in WG) * N_LAYERS


def weight_gradients(stream):
    for _ in range(N_LAYERS):
        for a, b, out in ops:
            lib.gemm_tn(a, b, out, accumulate=True, stream=stream)


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fa, fb, reps=5):
    """best wall time of fa on stream A and fb on stream B started together (either may be None)"""
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fb is not None:
            with torch.cuda.stream(sb):
                fb()
        if fa is not None:
            with torch.cuda.stream(sa):
                fa()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


print("wgrad under attention backward: %d decoder layers' attention backward (stream A) beside their weight gradients (stream B), "
      "64 segments x 1024 tokens, p = 0.1; ms, best of 5; library version %d" % (N_LAYERS, lib.load().mrmt3_version()))
for name, knob in (("round-1 128x128 TN tile kernel (MRMT3_TN8=0)", 0), ("ping-pong TN kernel (production, one launch per gradient)", 1)):
    lib.set_knob("MRMT3_TN8", knob)
    before = lib.dispatch_counts()
    for _ in range(2):                                    # warm-up (workspaces, code objects)
        wall(attention_backward, lambda: weight_gradients(sb), 1)
    after = lib.dispatch_counts()
    ta = wall(attention_backward, None)
    tb = wall(None, lambda: weight_gradients(sb))
    tab = wall(attention_backward, lambda: weight_gradients(sb))
    tba = wall(lambda: None, None)                         # (launch overhead of an empty round)
    kern = {k: after[k] - before[k] for k in ("tn_tile", "tn8", "attn_bwd", "attn_bwd_onepass") if after[k] != before[k]}
    print("  %-62s t(A) %7.3f   t(B) %7.3f (%.0f TFLOP/s)   t(A||B) %7.3f = %.3f x (t(A) + t(B))   [launches in the warm-up: %s]"
          % (name, ta, tb, flops_b / tb / 1e9, tab, tab / (ta + tb), kern))
lib.reset_knobs()
print("kill criterion: t(A||B) > 0.85 x (t(A) + t(B))")
