import sys, os, time
sys.path[:0] = ["/root/repo/mr-mt3_amd", "/root/repo"]
import torch
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5 import T5ForConditionalGeneration
dev = torch.device("cuda:0")
for p in (0.1, 0.0):
    m = T5ForConditionalGeneration(dict(T5_SMALL, dropout_rate=p)).load_golden().to(dev)
    tr = Trainer(m, lr=2e-4)
    a = torch.from_numpy(synth_audio(64, seed=365)).to(dev); l = torch.from_numpy(synth_labels(64, seed=365)).to(dev)
    for _ in range(6): tr.train_step(a, l, audio=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.train_step(a, l, audio=True)
    torch.cuda.synchronize(); print("dropout", p, "ms/step", (time.perf_counter() - t0) / 20 * 1e3)
    del tr, m; torch.cuda.empty_cache()
