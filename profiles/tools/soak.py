import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "mr-mt3_amd"))
import torch
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
dev = torch.device("cuda:0")
m = T5SegMemV2WithPrev(T5_SMALL, 1, 64).load_golden().to(dev)
tr = Trainer(m, lr=2e-4)
audio = torch.from_numpy(synth_audio(64, seed=1)).to(dev); lab = torch.from_numpy(synth_labels(64, seed=2)).to(dev); prev = torch.from_numpy(synth_labels(64, seed=3)).to(dev)
for i in range(400):
    loss = tr.train_step(audio, lab, prev.clone(), audio=True)
    if i in (5, 100, 399):
        torch.cuda.synchronize()
        print(i, "loss %.4f" % loss.item(), "alloc %.2f GiB reserved %.2f GiB" % (torch.cuda.memory_allocated()/2**30, torch.cuda.memory_reserved()/2**30), flush=True)
mel = torch.rand(8, 256, 512, device=dev)
for j in range(20):
    ids = m.generate(mel[:3], max_length=64)
print("decode ok", ids.shape, "reserved %.2f GiB" % (torch.cuda.memory_reserved()/2**30))
