"""Log-mel kernel: round-1 workgroup-per-frame kernel (MRMT3_LOGMEL=0) against round 4's wave-per-frame kernel, 64 / 12 / 1
segments of 2.048 s, bf16 and f32 output; us per launch (median of 30, back to back and cold) and the algorithmic HBM rate
(131 072 B in + 256 x 512 x {2,4} B out per segment).   python3 profiles/tools/logmel_micro.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from contrib import spectrograms as sp
from mrmt3.synthetic import synth_audio

dev = torch.device("cuda:0")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def t_us(fn, cold, reps=30):
    fn(); torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        if cold:
            flush.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in ev)
    return ts[len(ts) // 2] * 1e3


print("%-10s %-6s | %22s | %22s | %s" % ("segments", "out", "round-1 kernel warm/cold", "wave kernel warm/cold", "wave kernel cold: GB/s algorithmic, frac of 8 TB/s"))
for B in (64, 12, 1, 256):
    audio = torch.from_numpy(synth_audio(B, seed=3)).to(dev)
    for bf in (True, False):
        r = []
        for mode in ("0", "1"):
            os.environ["MRMT3_LOGMEL"] = mode
            fn = lambda: sp.logmel_segments(audio, out_bf16=bf)
            r += [t_us(fn, False), t_us(fn, True)]
        by = B * (131072 + 256 * 512 * (2 if bf else 4))
        print("%-10d %-6s | %10.1f %10.1f  | %10.1f %10.1f  | %8.0f  %.3f" % (B, "bf16" if bf else "f32", r[0], r[1], r[2], r[3], by / r[3] / 1e3, by / r[3] / 1e3 / 8000))
