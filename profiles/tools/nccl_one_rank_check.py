"""RCCL on the one GPU of the box: (1) the collectives the trainer uses work on slices of a flat buffer, (2) the
trainer's bucketed exchange (launch stream waiting for the backward stream and the wgrad side stream, async all-reduce,
wait before AdamW) runs over RCCL at world size 1 — a sum over one rank is the identity, so three training steps must
give bit-identical parameters with and without the forced collectives."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(1 << 20, device=dev)
w = dist.all_reduce(x[1000:500000], op=dist.ReduceOp.SUM, async_op=True)
w.wait()
dist.barrier()
dist.broadcast(x, src=0)
torch.cuda.synchronize()
print("nccl ok", x.sum().item(), dist.get_backend())

from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5 import T5ForConditionalGeneration


def run(force):
    os.environ["MRMT3_DDP_FORCE_COLLECTIVES"] = "1" if force else "0"
    m = T5ForConditionalGeneration(T5_SMALL).load_golden().to(dev)
    tr = Trainer(m, lr=2e-4)
    audio = torch.from_numpy(synth_audio(8, seed=1)).to(dev)
    labels = torch.from_numpy(synth_labels(8, seed=1)).to(dev)
    for _ in range(3):
        loss = tr.train_step(audio, labels, audio=True)
    torch.cuda.synchronize()
    return m.flat.P.clone(), float(loss)


p0, l0 = run(False)
p1, l1 = run(True)
print("forced collectives: loss", l1, "plain:", l0, "parameters identical:", bool(torch.equal(p0, p1)))
assert torch.equal(p0, p1)
dist.destroy_process_group()
