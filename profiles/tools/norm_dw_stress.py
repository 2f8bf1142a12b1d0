"""Stress of the batched norm-weight-gradient reduction (mrmt3_norm_dw_reduce: last-arriving chunk, relaxed agent-scope
atomics, no fence — ADVICE round 1) and of the deferred split-K slab reduce: N iterations on fixed inputs, every result
compared bit for bit with the first, alone and with a second PROCESS keeping the GPU busy (time slicing, which is what
the two-rank tests on one GPU do).   python3 profiles/tools/norm_dw_stress.py [iterations] [noise: 0/1]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
noise = len(sys.argv) > 2 and sys.argv[2] == "1"
child = None
if noise:      # started BEFORE this process touches the GPU
    code = ("import torch,time\nx=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)\nt=time.time()\n"
            "while time.time()-t<%d:\n  y=x@x\n  torch.cuda.synchronize()\n" % int(sys.argv[3] if len(sys.argv) > 3 else 60))
    child = subprocess.Popen([sys.executable, "-c", code])
    time.sleep(8)
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = {}
for rows in (256, 4096, 65536):
    n_sites = 20
    xs = [torch.randn(rows, 512, device=dev) for _ in range(n_sites)]
    gs = [torch.randn(rows, 512, device=dev).bfloat16() for _ in range(n_sites)]
    rstd = torch.rand(rows, device=dev) + 0.5
    w = torch.rand(512, device=dev) + 0.5
    dws = [torch.zeros(512, device=dev) for _ in range(n_sites)]
    batch = lib.NormDwBatch()
    a = torch.randn(rows, 512, device=dev).bfloat16()
    b = torch.randn(rows, 384, device=dev).bfloat16()
    tn = lib.TnBatch()
    gw = torch.zeros(512, 384, device=dev)
    first = None
    nbad = nbad_tn = 0
    for it in range(iters):
        for d in dws:
            d.zero_()
        gw.zero_()
        for i in range(n_sites):
            lib.add_rmsnorm_bwd(gs[i], None, xs[i], rstd, w, dws[i], want_dy=False, defer=batch)
        lib.gemm_tn(a, b, gw, accumulate=True, defer=tn)
        batch.flush()
        tn.flush()
        cur = torch.stack(dws).clone()
        if first is None:
            first, first_tn = cur, gw.clone()
            ref = torch.stack([((gs[i].float() * w) * 0 + gs[i].float() * xs[i] * rstd[:, None]).sum(0) for i in range(n_sites)])
            err = (first - ref).abs().max().item() / ref.abs().max().item()
            assert err < 1e-4, err
        else:
            nbad += int(not torch.equal(cur, first))
            nbad_tn += int(not torch.equal(gw, first_tn))
    torch.cuda.synchronize()
    bad[rows] = (nbad, nbad_tn)
    print("rows %6d: %d iterations, norm-dw results differing from the first: %d, slab-reduce: %d%s"
          % (rows, iters, nbad, nbad_tn, "  (second process active)" if noise else ""), flush=True)
if child is not None:
    child.wait()
