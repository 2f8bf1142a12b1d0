"""The data-parallel training step end to end on the GPU: two ranks (both on cuda:0 — one box has one GPU —
exchanging over gloo, which carries device tensors through the host; on the 8-GPU node the same code runs over
RCCL) against ONE process that sees both ranks' segments:

  * after a step both ranks hold bit-identical parameters (broadcast at start, same reduced gradients);
  * their reduced gradient equals the gradient of the mean loss over the global batch (each rank's loss is a mean
    over its own tokens, token counts equal), i.e. what the single process computes on the concatenated batch;
  * the side-stream weight gradients are joined before their bucket is exchanged (a missed join shows up as a
    mismatch here).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank, B=2):
    from mrmt3.synthetic import synth_audio, synth_labels
    return (torch.from_numpy(synth_audio(B, seed=50 + rank)), torch.from_numpy(synth_labels(B, 128, seed=60 + rank)))


def _model(dev):
    from mrmt3.synthetic import T5_SMALL
    from models.t5 import T5ForConditionalGeneration
    return T5ForConditionalGeneration(dict(T5_SMALL, dropout_rate=0.0)).load_golden().to(dev)


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "mr-mt3_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mrmt3.trainer import Trainer
        m = _model(dev)
        if rank == 1:
            with torch.no_grad():
                m.flat.P.mul_(1.5)                       # the trainer's initial broadcast must undo this
        tr = Trainer(m, lr=1e-3)
        audio, lab = _batch(rank)
        loss = tr.train_step(audio.to(dev), lab.to(dev), audio=True)
        torch.cuda.synchronize()
        q.put((rank, m.flat.G.cpu().numpy(), m.flat.P.cpu().numpy(), float(loss.item())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_one_process_on_the_global_batch():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, g0, p0, l0), (_, g1, p1, l1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)       # identical replicas after the step
    assert abs(l0 - l1) < 1e-6                                      # the logged loss is the all-reduced mean

    from mrmt3.trainer import Trainer
    dev = torch.device("cuda", 0)
    m = _model(dev)
    tr = Trainer(m, lr=1e-3)
    a0, t0 = _batch(0)
    a1, t1 = _batch(1)
    loss = tr.train_step(torch.cat([a0, a1]).to(dev), torch.cat([t0, t1]).to(dev), audio=True)
    torch.cuda.synchronize()
    g = m.flat.G.cpu().numpy()
    # reduced gradient = sum over ranks; the trainer folds 1/world into AdamW's grad_scale
    rel = np.linalg.norm(g0 / 2 - g) / np.linalg.norm(g)
    assert rel < 2e-2, rel                                          # bf16 GEMM rounding differs with the batch split
    assert abs(loss.item() - l0) < 2e-3
    dp = np.abs(p0 - m.flat.P.cpu().numpy()).max()
    assert dp < 2.5e-3, dp                                          # one AdamW step of lr 1e-3 moves a weight by <= ~1e-3
