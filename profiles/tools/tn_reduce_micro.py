"""The batched split-K slab reduction (mrmt3_tn_reduce_sites) on the weight-gradient sites of one training step:
bytes streamed, time, GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
dev = torch.device("cuda:0")
L = lib.load()
Md, Me = 65536, 16384
sites = []
for _ in range(8):
    sites += [(Md, 1152, 512), (Md, 512, 384), (Md, 384, 512), (Me, 768, 512), (Md, 512, 384), (Md, 2048, 512), (Md, 512, 1024)]
    sites += [(Me, 1152, 512), (Me, 512, 384), (Me, 2048, 512), (Me, 512, 1024)]
sites += [(Md, 1536, 512), (Me, 512, 512)]
batch = lib.TnBatch()
outs = [torch.zeros(n1, n2, device=dev) for _, n1, n2 in sites]
tot = 0
for (M, n1, n2), o in zip(sites, outs):
    buf = batch.site(o, M, n1, n2, True)
    buf.zero_()
    tot += L.mrmt3_gemm_tn_splits(M, n1, n2) * n1 * n2 * 4
keys = list(batch._queue)
def run():
    batch._queue[:] = keys
    batch.flush()
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(f"{len(sites)} sites, {tot/1e9:.2f} GB of slabs: {best*1e3:.0f} us = {tot/best/1e6:.0f} GB/s")
