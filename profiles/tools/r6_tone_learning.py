"""Round 6: how many steps does a T5-small need on the tone corpus of tests/test_trajectory_gpu.py before it transcribes held-out
segments?  bf16 + dropout 0.1 (the benchmark's arithmetic), B segments per step; prints the smoothed loss and the held-out onset
F1 (through InferenceHandler.inference) along the way.  usage: r6_tone_learning.py [steps] [batch] [lr]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mr-mt3_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import test_trajectory_gpu as T
from models.t5 import T5ForConditionalGeneration
from mrmt3.synthetic import T5_SMALL
from mrmt3.tokenizer import Tokenizer
from mrmt3.trainer import Trainer
from utils import cosine_warmup_lambda

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-4
dev = torch.device("cuda", 0)
tk = Tokenizer()
rs = np.random.RandomState(7)
m = T5ForConditionalGeneration(dict(T5_SMALL, dropout_rate=0.1)).load_golden().to(dev)
tr = Trainer(m, lr=lr, lr_lambda=cosine_warmup_lambda(20, max(4 * 300, steps), min_lr=1e-4))
losses, t0 = [], time.time()
for i in range(1, steps + 1):
    a, lab, _ = T.tone_batch(rs, B, tk)
    losses.append(tr.train_step(torch.from_numpy(a).to(dev), lab.to(dev), audio=True))
    if i % 250 == 0 or i == steps:
        vals = [float(x.item()) for x in losses[-100:]]
        line = "step %5d  %.0f s  loss (median of last 100) %.4f  min %.4f" % (i, time.time() - t0, float(np.median(vals)), min(vals))
        if i % 500 == 0 or i == steps:
            f1, p, r = T.onset_f1(dev, m, 16)
            m.train()
            line += "   held-out onset F1 %.3f (P %.3f R %.3f)" % (f1, p, r)
        print(line, flush=True)
