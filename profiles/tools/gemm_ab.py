"""A/B of the two NT GEMM kernels (gemm.hip: one barrier per K step; gemm8.hip: ping-pong phases) on every NT shape of
the training step, in ONE process, interleaved rounds (cdna_hip_programming.md rule 24), random operands, with a
correctness check of the new kernel against an f32 torch product on sampled rows.
Usage: python profiles/tools/gemm_ab.py [rounds] [reps] [segments per GPU = 64] [cold]
`cold`: every timed launch is preceded by a 512 MiB write (the Infinity Cache no longer holds the operands, as inside
the training step, where ~100 GB pass between two uses of a weight) and timed on its own with events — the
back-to-back loop of the default mode re-reads operands that sit in the 256 MB cache and flatters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
lib.load()
os.environ.setdefault("MRMT3_GEMM8_ALL", "1")      # A/B every admissible shape, not only the ones the dispatch rule takes
SEG = int(sys.argv[3]) if len(sys.argv) > 3 else 64
COLD = len(sys.argv) > 4 and sys.argv[4] == "cold"
os.environ.setdefault("MRMT3_GEMM8_MIN_M", "1024")  # (tuning) let the ping-pong kernel take the short encoder shapes too
Md, Me = SEG * 1024, SEG * 256
flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if COLD else None
# (name, M, N, K, out dtype, launches per step)
NT = [("qkv", Md, 1152, 512, "bf16", 8), ("o/co", Md, 512, 384, "bf16", 16), ("cq", Md, 384, 512, "bf16", 8),
      ("ckv_all", Me, 6144, 512, "bf16", 1), ("wi", Md, 2048, 512, "bf16", 8), ("wo", Md, 512, 1024, "bf16", 8),
      ("lm_head", Md, 1536, 512, "f32", 1), ("d_lm", Md, 512, 1536, "f32", 1),
      ("d_qkv", Md, 512, 1152, "bf16", 8), ("d_wi", Md, 512, 2048, "bf16", 8), ("d_wo", Md, 1024, 512, "bf16", 8),
      ("d_o/co", Md, 384, 512, "bf16", 16), ("d_cq", Md, 512, 384, "bf16", 8), ("d_ckv_all", Me, 512, 6144, "f32", 1),
      ("e_qkv", Me, 1152, 512, "bf16", 8), ("e_o", Me, 512, 384, "bf16", 8), ("e_wi", Me, 2048, 512, "bf16", 8),
      ("e_wo", Me, 512, 1024, "bf16", 8), ("e_dqkv", Me, 512, 1152, "bf16", 8), ("e_dwi", Me, 512, 2048, "bf16", 8),
      ("e_dwo", Me, 1024, 512, "bf16", 8), ("e_do", Me, 384, 512, "bf16", 8)]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    if COLD:
        evs = []
        for _ in range(reps):
            flush_buf.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2] * 1e-3                      # median of the individually timed cold launches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


tot = {"0": 0.0, "1": 0.0}
flops = 0.0
for name, M, N, K, od, per_step in NT:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    acc = od == "f32+"
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16 if od == "bf16" else torch.float32)
    # correctness of the new kernel on sampled rows (and the masked / overlapping last column tile)
    os.environ["MRMT3_GEMM8"] = "1"
    out.zero_()
    lib.gemm_nt(a, b, out=out, accumulate=acc)
    rows = torch.randint(0, M, (min(512, M),), device=dev)
    ref = a[rows].float() @ b.float().t()
    err = (out[rows].float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (1e-2 if od == "bf16" else 1e-5), (name, err)
    if acc:
        lib.gemm_nt(a, b, out=out, accumulate=True)
        assert (out[rows] - 2 * ref).abs().max().item() / ref.abs().max().item() < 1e-5
    best = {"0": 1e9, "1": 1e9}
    for _ in range(rounds):
        for k in ("0", "1"):
            os.environ["MRMT3_GEMM8"] = k
            best[k] = min(best[k], timeit(lambda: lib.gemm_nt(a, b, out=out, accumulate=acc)))
    f = 2.0 * M * N * K
    for k in best:
        tot[k] += best[k] * per_step
    flops += f * per_step
    print(f"NT {name:8s} M={M:5d} N={N:4d} K={K:4d} {od:5s}: old {best['0']*1e6:7.1f} us {f/best['0']/1e12:6.0f} TF | "
          f"new {best['1']*1e6:7.1f} us {f/best['1']/1e12:6.0f} TF | x{best['0']/best['1']:.2f}  (max rel err {err:.1e})")
print(f"[{SEG} segments per GPU, {'cold (512 MiB written before every launch)' if COLD else 'back to back'}] per step: old {tot['0']*1e3:.2f} ms ({flops/tot['0']/1e12:.0f} TF), new {tot['1']*1e3:.2f} ms ({flops/tot['1']/1e12:.0f} TF)")
