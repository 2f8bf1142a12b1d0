"""Micro-benchmark of the GEMM shapes the training step launches (batch 64).  Prints TFLOP/s and the
HBM-traffic floor per shape.  Usage: python profiles/tools/gemm_micro.py [nt|tn|all] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
if os.environ.get("MRMT3_TOOL_LIB"):      # tuning tool only: A/B a variant build of the library
    lib.LIB_PATH = os.environ["MRMT3_TOOL_LIB"]

which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
lib.load()
Md, Me = 65536, 16384
NT = [("qkv", Md, 1152, 512, "bf16"), ("o", Md, 512, 384, "f32"), ("cq", Md, 384, 512, "bf16"),
      ("ckv", Me, 768, 512, "bf16"), ("wi", Md, 2048, 512, "bf16"), ("wo", Md, 512, 1024, "f32"),
      ("lm_head", Md, 1536, 512, "f32"), ("d_qkv", Md, 512, 1152, "f32"), ("d_wi", Md, 512, 2048, "f32"),
      ("d_wo", Md, 1024, 512, "bf16"), ("d_o", Md, 384, 512, "bf16"),
      ("e_qkv", Me, 1152, 512, "bf16"), ("e_o", Me, 512, 384, "f32"), ("e_wi", Me, 2048, 512, "bf16"),
      ("e_wo", Me, 512, 1024, "f32"), ("e_dqkv", Me, 512, 1152, "f32"), ("e_dwi", Me, 512, 2048, "f32"),
      ("e_dwo", Me, 1024, 512, "bf16"), ("e_do", Me, 384, 512, "bf16")]
TN = [("w_qkv", Md, 1152, 512), ("w_o", Md, 512, 384), ("w_cq", Md, 384, 512), ("w_wi", Md, 2048, 512),
      ("w_wo", Md, 512, 1024), ("w_lm", Md, 1536, 512), ("w_ckv", Me, 768, 512), ("e_qkv", Me, 1152, 512),
      ("e_o", Me, 512, 384), ("e_wi", Me, 2048, 512), ("e_wo", Me, 512, 1024)]
if os.environ.get("MRMT3_TN_SWAPPED"):    # orientation study: the same products with the operands exchanged
    TN = [(n + "^T", M, b, a) for n, M, a, b in TN]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


if which in ("nt", "all"):
    for name, M, N, K, od in NT:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if od == "bf16" else torch.float32)
        t = timeit(lambda: lib.gemm_nt(a, b, out=out))
        byts = a.numel() * 2 + b.numel() * 2 + out.numel() * out.element_size()
        print(f"NT {name:8s} M={M} N={N} K={K} out={od}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  "
              f"{byts/t/1e9:7.0f} GB/s (floor {byts/5e12*1e6:6.1f} us @5TB/s)")
if which in ("tn", "all"):
    for name, M, N1, N2 in TN:
        a = torch.randn(M, N1, device=dev).bfloat16()
        b = torch.randn(M, N2, device=dev).bfloat16()
        out = torch.zeros(N1, N2, device=dev)
        t = timeit(lambda: lib.gemm_tn(a, b, out, accumulate=True))
        byts = a.numel() * 2 + b.numel() * 2
        print(f"TN {name:8s} M={M} N1={N1} N2={N2}: {t*1e6:8.1f} us  {2*M*N1*N2/t/1e12:7.1f} TF/s  {byts/t/1e9:7.0f} GB/s")
