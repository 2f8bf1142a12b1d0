"""Per-phase shader-clock ticks of the attention forward loop (needs the ATTN_XP=5 tuning build of the library:
hipcc ... -DATTN_XP=5 -c attention.hip; MRMT3_TOOL_LIB=<that .so> python profiles/tools/attn_phases.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
lib.LIB_PATH = os.environ["MRMT3_TOOL_LIB"]
dev = torch.device("cuda:0")
lib.load()
dbg = ctypes.CDLL(lib.LIB_PATH).mrmt3_dbg_attn
B, H = 64, 6
names = ["wait vmcnt", "barrier", "stage + K reads + S mfma issue", "softmax (incl. S wait)", "PV (V reads + mfma)", "whole loop"]
for name, Lq, Lk, causal in (("dec-self", 1024, 1024, True), ("dec-cross", 1024, 256, False), ("enc-self", 256, 256, False)):
    qkv = torch.randn(B * Lq, 1152, device=dev).bfloat16()
    qkv[:, :384] *= 0.35
    kv = torch.randn(B * Lk, 768, device=dev).bfloat16()
    q = qkv[:, :384]
    k, v = (qkv[:, 384:768], qkv[:, 768:]) if Lq == Lk else (kv[:, :384], kv[:, 384:])
    for p in (0.0, 0.1):
        lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=1, stream_id=1)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        dbg(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=1, stream_id=1)
        e1.record()
        torch.cuda.synchronize()
        dbg(buf, 1)
        t = list(buf)
        waves, tiles = t[7], t[6]
        print(f"{name} p={p}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {waves} waves, {tiles/waves:.1f} tiles/wave, "
              f"loop ticks/wave-tile {t[5]/tiles:.0f}")
        for i in range(5):
            print(f"    {names[i]:34s} {t[i]/tiles:8.0f} ticks/wave-tile  {100.0*t[i]/t[5]:5.1f}%")
