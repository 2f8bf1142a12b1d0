"""A/B of the two weight-gradient (TN) GEMM kernels (gemm.hip: 128x128 tiles, 2 workgroups per CU; gemm_tn8.hip:
256x256 ping-pong) on every TN shape of the training step, in one process, interleaved rounds, random operands, with a
check against an f32 torch product.  Times the MFMA kernel alone (deferred form: slabs only) and the immediate form
(kernel + its own slab reduce).  Usage: python profiles/tools/gemm_tn_ab.py [rounds] [reps]"""
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
os.environ.setdefault("MRMT3_TN8_ALL", "1")       # A/B every admissible shape, not only the ones the dispatch rule takes
Md, Me = 65536, 16384
TN = [("w_qkv", Md, 1152, 512, 8), ("w_o/co", Md, 512, 384, 16), ("w_cq", Md, 384, 512, 8), ("w_wi", Md, 2048, 512, 8),
      ("w_wo", Md, 512, 1024, 8), ("w_lm", Md, 1536, 512, 1), ("w_ckv", Me, 768, 512, 8), ("e_qkv", Me, 1152, 512, 8),
      ("e_o", Me, 512, 384, 8), ("e_wi", Me, 2048, 512, 8), ("e_wo", Me, 512, 1024, 8), ("e_proj", Me, 512, 512, 1)]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


tot = {"0": 0.0, "1": 0.0}
flops = 0.0
for name, M, N1, N2, per_step in TN:
    a = (torch.randn(M, N1, device=dev) * 0.1).bfloat16()
    b = (torch.randn(M, N2, device=dev) * 0.1).bfloat16()
    ref = (a[:8192].float().t() @ b[:8192].float())
    for k in ("0", "1"):
        os.environ["MRMT3_TN8"] = k
        out = torch.zeros(N1, N2, device=dev)
        lib.gemm_tn(a[:8192], b[:8192], out)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, (name, k, err)
    full = torch.zeros(N1, N2, device=dev)
    os.environ["MRMT3_TN8"] = "0"
    lib.gemm_tn(a, b, full)
    got = torch.zeros(N1, N2, device=dev)
    os.environ["MRMT3_TN8"] = "1"
    lib.gemm_tn(a, b, got)
    rel = ((got - full).abs().max() / full.abs().max()).item()
    assert rel < 2e-5, (name, rel)
    best = {"0": 1e9, "1": 1e9}
    best_i = {"0": 1e9, "1": 1e9}
    batches = {"0": lib.TnBatch(), "1": lib.TnBatch()}
    for _ in range(rounds):
        for k in ("0", "1"):
            os.environ["MRMT3_TN8"] = k
            bt = batches[k]

            def part():
                lib.gemm_tn(a, b, got, accumulate=True, defer=bt)
                bt._queue.clear()
            best[k] = min(best[k], timeit(part))
            best_i[k] = min(best_i[k], timeit(lambda: lib.gemm_tn(a, b, got, accumulate=False)))
    f = 2.0 * M * N1 * N2
    for k in best:
        tot[k] += best[k] * per_step
    flops += f * per_step
    print(f"TN {name:7s} M={M:5d} N1={N1:4d} N2={N2:4d}: old {best['0']*1e6:7.1f} us {f/best['0']/1e12:5.0f} TF (+reduce {best_i['0']*1e6:7.1f}) | "
          f"new {best['1']*1e6:7.1f} us {f/best['1']/1e12:5.0f} TF (+reduce {best_i['1']*1e6:7.1f}) | x{best['0']/best['1']:.2f}")
print(f"per step (MFMA kernels only): old {tot['0']*1e3:.2f} ms ({flops/tot['0']/1e12:.0f} TF), new {tot['1']*1e3:.2f} ms ({flops/tot['1']/1e12:.0f} TF)")
