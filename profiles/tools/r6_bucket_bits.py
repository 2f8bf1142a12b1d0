"""Round 6: does the gradient of a step depend on HOW the backward is driven?  Same weights, same batch (16 x 256 tokens: the
shapes where the grouped weight-gradient launch, the fused row kernels and the ping-pong GEMM dispatch), MR-MT3's own model:
  a  engine.backward(tape, dl)                         first backward of the process
  b  the same again                                    (warm: tables, workspaces exist)
  c  engine.backward(tape, dl, on_layer_done=...)      buckets fired mid-backward (join_wgrad at every bucket boundary), world 1
Prints, per parameter group of the flat buffer, the largest |difference| a-b and b-c."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mr-mt3_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from contrib import spectrograms as sp
from mrmt3 import lib
from mrmt3.ddp import layer_ranges
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev

dev = torch.device("cuda", 0)
B, L = 16, 256
mel = sp.logmel_segments(torch.from_numpy(synth_audio(B, seed=50)).to(dev), out_bf16=True)
lab = torch.from_numpy(synth_labels(B, L, seed=60)).to(dev)
prev = torch.from_numpy(synth_labels(B, L, seed=70, full=False, mean_len=L // 2)).to(dev)
m = T5SegMemV2WithPrev(dict(T5_SMALL, dropout_rate=0.0), 1, 64).load_golden().to(dev)
tr = Trainer(m, lr=1e-3, graph=False)
eng, flat = tr.engine, tr.flat
m.train()


def grad(on_layer_done=None, fire=False):
    eng.reset_deferred()
    eng._stream_ctr = 0
    dec, tape = eng.forward(mel, lab, prev.clone(), training=True, need_grad=True, want_logits=False)
    loss, dl = lib.lmhead_cross_entropy(dec, eng.W("lm_head"), lab.reshape(-1), want_grad=True, grad_dtype=torch.bfloat16)
    flat.G.zero_()
    if fire:
        sent = []

        def done(prefix, i):
            idx = tr.buckets.triggered_by(prefix, i)
            if idx:
                eng.join_wgrad()
                sent.extend(idx)
        eng.backward(tape, dl, on_layer_done=done)
        print("   buckets completed in order:", sent)
    else:
        eng.backward(tape, dl)
    torch.cuda.synchronize()
    return flat.G.clone(), float(loss.item())


lib.dispatch_counts(reset=True)
ga, la = grad()
print("a counts:", {k: v for k, v in lib.dispatch_counts(reset=True).items() if v})
gb, lb = grad()
print("b counts:", {k: v for k, v in lib.dispatch_counts(reset=True).items() if v})
gc_, lc = grad(fire=True)
print("c counts:", {k: v for k, v in lib.dispatch_counts(reset=True).items() if v})
gd, ld = grad()
print("losses", la, lb, lc, ld)
for name, x, y in (("a-b (first vs second backward)", ga, gb), ("b-c (plain vs joined at bucket boundaries)", gb, gc_),
                   ("b-d (plain, repeated)", gb, gd)):
    d = (x - y).abs()
    print("%s: max |diff| %.3e of max |g| %.3e, %d of %d elements differ" % (name, float(d.max()), float(x.abs().max()),
                                                                             int((d > 0).sum()), d.numel()))
    if float(d.max()) > 0:
        for tag, a, b in layer_ranges(flat):
            dm = float(d[a:b].max())
            if dm > 0:
                print("    %-22s max |diff| %.3e  max |g| %.3e  differing %d / %d" % (tag, dm, float(x[a:b].abs().max()),
                                                                                     int((d[a:b] > 0).sum()), b - a))
