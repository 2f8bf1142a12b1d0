"""Does any kernel of the training step consume memory it (or its producer) never wrote?  Fill the caching allocator's
free blocks with NaN (or a large finite value) before the step and compare the gradients with an unpoisoned run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import numpy as np, torch
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from models.t5 import T5ForConditionalGeneration
from mrmt3.trainer import Trainer
dev = torch.device("cuda:0")


def poison(val):
    keep = []
    for sz in [2 ** k for k in range(9, 29)] + [3 * 2 ** k for k in range(9, 27)] + [5 * 2 ** k for k in range(9, 26)]:
        for _ in range(3 if sz < (1 << 24) else 1):
            t = torch.empty(sz // 4, device=dev, dtype=torch.float32)
            t.fill_(val)
            keep.append(t)
    torch.cuda.synchronize()
    del keep


def run(val, B, L, steps=2):
    torch.cuda.empty_cache()
    if val is not None:
        poison(val)
    m = T5ForConditionalGeneration(dict(T5_SMALL, dropout_rate=0.0)).load_golden().to(dev)
    tr = Trainer(m, lr=1e-3, graph=False)
    a = torch.from_numpy(synth_audio(B, seed=50)).to(dev)
    t = torch.from_numpy(synth_labels(B, L, seed=60)).to(dev)
    for _ in range(steps):
        if val is not None:
            poison(val)
        loss = tr.train_step(a, t, audio=True)
    torch.cuda.synchronize()
    return loss.item(), m.flat.G.cpu().numpy().copy(), m.flat.P.cpu().numpy().copy()


for B, L in ((2, 128), (3, 192), (8, 256)):
    l0, g0, p0 = run(None, B, L)
    for val in (float("nan"), 1e4):
        l1, g1, p1 = run(val, B, L)
        bad = ~np.isfinite(g1)
        d = np.abs(np.nan_to_num(g1) - g0)
        print("B=%d L=%d poison=%s: loss %.6f vs %.6f, non-finite grads %d, differing grads %d (max %.3e), P differing %d"
              % (B, L, val, l0, l1, int(bad.sum()), int((d > 0).sum()), d.max(), int((p0 != p1).sum())), flush=True)
        if (d > 0).any():
            from mrmt3.params import FlatParams
            f = FlatParams(T5_SMALL)
            idx = np.nonzero(d)[0]
            offs = list(f.offsets.items())
            hit = {}
            for i in idx[:: max(1, len(idx) // 2000)]:
                for (k, o), nxt in zip(offs, offs[1:] + [(None, f.numel)]):
                    if o <= i < nxt[1]:
                        hit[k] = hit.get(k, 0) + 1
                        break
            print("   in:", hit, flush=True)
