"""A/B of the fused "projection + row kernel" launches (csrc/gemm_rows.hip) against the two kernels they replace, per site
shape of the training step, COLD (512 MiB written before every timed launch, as inside the step) and warm (back to back),
one process, interleaved.
    python3 profiles/tools/gemm_rows_ab.py [segments per GPU = 64] [reps = 12]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

SEG = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
lib.load()
Md, Me = SEG * 1024, SEG * 256
flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, cold):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        if cold:
            flush_buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3          # us


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


rows_out = []
# ---- forward: projection -> add + norm
for name, M, K, per_step in (("o/co", Md, 384, 16), ("wo", Md, 1024, 8), ("e_o", Me, 384, 8), ("e_wo", Me, 1024, 8)):
    a, w = rnd(M, K), rnd(512, K, scale=K ** -0.5)
    x0 = rnd(M, 512, dtype=torch.float32)
    wn = torch.ones(512, device=dev)
    x1 = torch.empty_like(x0)
    y = torch.empty(M, 512, device=dev, dtype=torch.bfloat16)

    def two():
        lib.gemm_nt(a, w, out=y)
        lib.add_rmsnorm_fwd(x0, y, wn, 1e-6, torch.bfloat16, p=0.1, seed=1, stream_y=3, x1=x1)

    def gemm_only():
        lib.gemm_nt(a, w, out=y)

    def fused():
        lib.gemm_nt_addnorm(a, w, x0, wn, 1e-6, p=0.1, seed=1, stream_y=3, x1=x1)

    r = [name, M, K, per_step]
    for cold in (True, False):
        r += [timeit(two, cold), timeit(gemm_only, cold), timeit(fused, cold)]
    rows_out.append(("addnorm", r, 2.0 * M * 512 * K, (M * K + 512 * K) * 2 + M * 512 * 10))
# ---- backward: data gradient -> norm backward
for name, M, K, per_step in (("d_qkv", Md, 1152, 8), ("d_cq", Md, 384, 8), ("d_wi", Md, 2048, 8), ("e_dqkv", Me, 1152, 8),
                             ("e_dwi", Me, 2048, 8)):
    a, wt = rnd(M, K), rnd(512, K, scale=K ** -0.5)
    dres = rnd(M, 512)
    x1 = rnd(M, 512, dtype=torch.float32)
    rstd = torch.rsqrt((x1 * x1).mean(-1) + 1e-6)
    wn = torch.ones(512, device=dev)
    dw = torch.zeros(512, device=dev)
    batch = lib.NormDwBatch()
    dxn = torch.empty(M, 512, device=dev, dtype=torch.bfloat16)
    dx1 = torch.empty(M, 512, device=dev, dtype=torch.bfloat16)

    def two():
        lib.gemm_nt(a, wt, out=dxn)
        lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, wn, dw, p=0.1, seed=1, stream_y=3, dx1=dx1, defer=batch)
        batch._queue.clear()

    def gemm_only():
        lib.gemm_nt(a, wt, out=dxn)

    def fused():
        lib.gemm_nt_normbwd(a, wt, dres, x1, rstd, wn, dw, p=0.1, seed=1, stream_y=3, dx1=dx1, defer=batch)
        batch._queue.clear()

    r = [name, M, K, per_step]
    for cold in (True, False):
        r += [timeit(two, cold), timeit(gemm_only, cold), timeit(fused, cold)]
    rows_out.append(("normbwd", r, 2.0 * M * 512 * K, (M * K + 512 * K) * 2 + M * 512 * 10))
# ---- backward: wo data gradient -> GEGLU backward
for name, M, per_step in (("d_wo", Md, 8), ("e_dwo", Me, 8)):
    dy, wt = rnd(M, 512), rnd(1024, 512, scale=512 ** -0.5)
    h = rnd(M, 2048)
    dg = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)

    def two():
        lib.gemm_nt(dy, wt, out=dg)
        lib.geglu_bwd(h, dg, p=0.1, seed=1, stream_id=3)

    def gemm_only():
        lib.gemm_nt(dy, wt, out=dg)

    def fused():
        lib.gemm_nt_geglubwd(dy, wt, h, p=0.1, seed=1, stream_id=3)

    r = [name, M, 512, per_step]
    for cold in (True, False):
        r += [timeit(two, cold), timeit(gemm_only, cold), timeit(fused, cold)]
    rows_out.append(("geglubwd", r, 2.0 * M * 1024 * 512, (M * 512 + 1024 * 512) * 2 + M * 1024 * 8))

print("%d segments per GPU; us per launch, median of %d; cold = 512 MiB written before every launch" % (SEG, reps))
print("%-9s %-7s %6s %5s | %28s | %28s | %s" % ("epilogue", "site", "rows", "K", "cold: two / product / fused", "warm: two / product / fused",
                                                 "fused cold: TFLOP/s, GB/s of algorithmic bytes"))
tot_two = tot_f = 0.0
for kind, r, fl, by in rows_out:
    name, M, K, per = r[:4]
    c2, cg, cf, w2, wg, wf = r[4:]
    tot_two += per * c2
    tot_f += per * cf
    print("%-9s %-7s %6d %5d | %8.1f %8.1f %8.1f   | %8.1f %8.1f %8.1f   | %6.0f %6.0f" %
          (kind, name, M, K, c2, cg, cf, w2, wg, wf, fl / cf / 1e6, by / cf / 1e3))
print("per step (cold sums, launches per step as listed): two kernels %.2f ms, fused %.2f ms" % (tot_two / 1e3, tot_f / 1e3))
