"""BASELINE configs[4] shapes: segmem_v2_with_prev, 2048-frame segments (+64 memory slots), 1024-token targets.
   python profiles/tools/long_context_step.py [batch] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
m = T5SegMemV2WithPrev(T5_SMALL, 1, 64, compute_dtype=torch.bfloat16).load_golden().to(dev)
tr = Trainer(m, lr=1e-5)
audio = torch.from_numpy(synth_audio(B, 2048 * 128, seed=1)).to(dev)
lab = torch.from_numpy(synth_labels(B, seed=2)).to(dev)
prev = torch.from_numpy(synth_labels(B, seed=3)).to(dev)
for _ in range(2):
    tr.train_step(audio, lab, prev.clone(), audio=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.train_step(audio, lab, prev.clone(), audio=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"long context B={B}: {dt*1e3:.2f} ms/step, {B/dt:.1f} segments/s (16.4 s of audio each), loss {loss.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
