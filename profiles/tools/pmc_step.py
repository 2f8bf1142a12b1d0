"""The training step of the benchmark (MT3Net, 64 segments, bf16, dropout on) launched eagerly a few times: the subject
of the whole-step PMC passes of profiles/tools/pmc_step_traffic.sh (FETCH_SIZE / WRITE_SIZE per kernel).
    python3 profiles/tools/pmc_step.py [steps = 3] [segments = 64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5 import T5ForConditionalGeneration

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
m = T5ForConditionalGeneration(T5_SMALL).load_golden().to(dev)
m.engine.overlap_wgrad = False                      # one stream, the chain the replayed graph runs
tr = Trainer(m, lr=2e-4, graph=False)
audio = torch.from_numpy(synth_audio(B, seed=365)).to(dev)
labels = torch.from_numpy(synth_labels(B, seed=365)).to(dev)
for _ in range(steps):
    loss = tr.train_step(audio, labels, audio=True)
torch.cuda.synchronize()
print("steps", steps, "segments", B, "loss %.4f" % loss.item())
