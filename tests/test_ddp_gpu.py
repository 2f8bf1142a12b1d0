"""The data-parallel training step end to end on the GPU: two ranks (both on cuda:0 — one box has one GPU —
exchanging over gloo, which carries device tensors through the host; on the 8-GPU node the same code runs over
RCCL) against ONE process that sees both ranks' segments:

  * after a step both ranks hold bit-identical parameters (broadcast at start, same reduced gradients);
  * their reduced gradient equals the gradient of the mean loss over the global batch (each rank's loss is a mean
    over its own tokens, token counts equal), i.e. what the single process computes on the concatenated batch;
  * the side-stream weight gradients are joined before their bucket is exchanged (a missed join shows up as a
    mismatch here).

Most tests here feed the ranks the log-mel INPUT, computed once in this (parent) process: rounds 2-3 saw "1 in 40" runs diverge
when two processes shared the GPU and traced it to the log-mel kernel coming out wrong beside the other process's LDS-DMA +
MFMA kernels (profiles/r03_two_process_soak.txt).  Round 4 found the cause — packed f32 VALU instructions emitted by the
compiler's SLP vectoriser misbehave when such kernels share the CU, from another stream of one process as well
(profiles/r04_lds_read_fault.txt) — and builds the library without them; test_two_ranks_with_the_frontend_inside_the_step…
now runs the once-failing configuration (raw audio in, both ranks on one GPU) and asserts bitwise repeatability.
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


def _batch(rank, B=2, Ld=128, prev=False):
    """(log-mel [B, 256, 512] bf16 as a CPU tensor, labels[, targets_prev]): the mel is made here, in the calling process, on
    the GPU.  prev=True adds the `targets_prev` tensor of MR-MT3's own model (dataset_2_random_prev_augment.py:61-74): ragged,
    -100 padded, from a stream of its own."""
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio, synth_labels
    audio = torch.from_numpy(synth_audio(B, seed=50 + rank)).cuda()
    mel = sp.logmel_segments(audio, out_bf16=True)
    out = (mel.cpu(), torch.from_numpy(synth_labels(B, Ld, seed=60 + rank)))
    if prev:
        out += (torch.from_numpy(synth_labels(B, Ld, seed=70 + rank, full=False, mean_len=Ld // 2)),)
    return out


def _model(dev, variant="t5"):
    from mrmt3.synthetic import T5_SMALL
    cfg = dict(T5_SMALL, dropout_rate=0.0)
    if variant == "t5":
        from models.t5 import T5ForConditionalGeneration
        return T5ForConditionalGeneration(cfg).load_golden().to(dev)
    if variant == "v2":
        from models.t5_segmem_v2 import T5SegMemV2
        return T5SegMemV2(cfg, 1, 64).load_golden().to(dev)
    assert variant == "with_prev", variant
    from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
    return T5SegMemV2WithPrev(cfg, 1, 64).load_golden().to(dev)


def _worker(rank, world, port, q, batch, steps=1, graph=False, audio=False, variant="t5"):
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
        m = _model(dev, variant)
        if rank == 1:
            with torch.no_grad():
                m.flat.P.mul_(1.5)                       # the trainer's initial broadcast must undo this
        tr = Trainer(m, lr=1e-3, graph=graph)
        if variant != "t5":
            # the memory encoder's gradients leave MID-backward, between the decoder's and the encoder's (ddp.py, engine.py)
            trig = [b["trigger"] for b in tr.buckets.buckets]
            assert len(trig) == 6 and ("segmem", 0) in trig and trig.index(("segmem", 0)) == 5, trig
            seg = tr.buckets.buckets[5]
            assert all(t.startswith("segmem") for t in seg["tags"]) and seg["end"] - seg["start"] == 2622976, seg
        mel, lab = batch[:2]
        prev = batch[2] if len(batch) > 2 else None
        from mrmt3 import lib
        lib.dispatch_counts(reset=True)
        for _ in range(steps):
            loss = tr.train_step(mel.to(dev), lab.to(dev), None if prev is None else prev.to(dev), audio=audio)
        torch.cuda.synchronize()
        counts = lib.dispatch_counts()
        assert tr.graph_captured == (graph and steps > 2)
        if tr.graph_captured:       # one graph per gradient bucket, the collectives stay eager between the replays
            cap = next(iter(tr._graphs.values()))
            assert len(cap.segments) == len(tr.buckets.buckets) >= 4
        q.put((rank, m.flat.G.cpu().numpy(), m.flat.P.cpu().numpy(), float(loss.item()), counts, len(tr.buckets.buckets)))
    finally:
        dist.destroy_process_group()


def _audio_batch(rank, B=2, Ld=128):
    """(raw audio [B, 32768] f32 as a CPU tensor, labels): the ranks run the log-mel frontend themselves."""
    from mrmt3.synthetic import synth_audio, synth_labels
    return torch.from_numpy(synth_audio(B, seed=50 + rank)), torch.from_numpy(synth_labels(B, Ld, seed=60 + rank))


def _run_two_ranks(steps=1, graph=False, B=2, Ld=128, with_counts=False, audio=False, variant="t5", batches=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    # (audio=False: log-mel computed here, before the ranks share the GPU — how rounds 3-4 kept the frontend out of the
    # two-process region while the co-residency fault was open)
    if batches is None:
        batches = [(_audio_batch if audio else _batch)(r, B, Ld) for r in range(2)]
    torch.cuda.synchronize()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, batches[r], steps, graph, audio, variant)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res if with_counts else [r[:4] for r in res]


def test_two_ranks_segmented_graph_replay_equals_eager():
    """World 2: the step is captured as one hipGraph per gradient bucket with the (gloo here, RCCL on the node)
    all-reduces enqueued eagerly between the replays.  Five steps (2 eager, capture, 2 replays) land on the same
    bits as five eager steps, on both ranks."""
    assert torch.cuda.is_available()

    def same(a, b):
        return all(r0 == r1 and np.array_equal(g0, g1) and np.array_equal(p0, p1) and abs(l0 - l1) < 2e-6
                   for (r0, g0, p0, l0), (r1, g1, p1, l1) in zip(a, b))
    eager = _run_two_ranks(steps=5, graph=False)
    graph = _run_two_ranks(steps=5, graph=True)
    assert same(eager, graph)                  # on the first try (round 2 retried here: see the module docstring)
    assert np.array_equal(graph[0][2], graph[1][2])


def test_two_ranks_match_one_process_on_the_global_batch():
    assert torch.cuda.is_available()
    res = _run_two_ranks()
    (_, g0, p0, l0), (_, g1, p1, l1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)       # identical replicas after the step
    assert abs(l0 - l1) < 1e-6                                      # the logged loss is the all-reduced mean

    from mrmt3.trainer import Trainer
    dev = torch.device("cuda", 0)
    m = _model(dev)
    tr = Trainer(m, lr=1e-3, graph=False)
    a0, t0 = _batch(0)
    a1, t1 = _batch(1)
    loss = tr.train_step(torch.cat([a0, a1]).to(dev), torch.cat([t0, t1]).to(dev), audio=False)
    torch.cuda.synchronize()
    g = m.flat.G.cpu().numpy()
    # reduced gradient = sum over ranks; the trainer folds 1/world into AdamW's grad_scale
    rel = np.linalg.norm(g0 / 2 - g) / np.linalg.norm(g)
    assert rel < 2e-2, rel                                          # bf16 GEMM rounding differs with the batch split
    assert abs(loss.item() - l0) < 2e-3
    dp = np.abs(p0 - m.flat.P.cpu().numpy()).max()
    assert dp < 2.5e-3, dp                                          # one AdamW step of lr 1e-3 moves a weight by <= ~1e-3


def test_two_ranks_with_the_frontend_inside_the_step_are_repeatable():
    """The configuration that diverged "1 in 20" in rounds 2-3: two processes on one GPU, raw AUDIO in, the log-mel kernel
    running beside the other rank's flash-attention / tile kernels.  Round 4 traced the fault to packed f32 VALU
    instructions (profiles/r04_lds_read_fault.txt) and builds the library without them: three independent two-rank runs of
    five steps (eager, then graph-replayed) land on the same bits, and both ranks hold identical replicas."""
    assert torch.cuda.is_available()
    runs = [_run_two_ranks(steps=5, graph=(i == 2), audio=True) for i in range(3)]
    for res in runs:
        (_, g0, p0, l0), (_, g1, p1, l1) = res
        assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    for res in runs[1:]:
        for (r0, g0, p0, l0), (r1, g1, p1, l1) in zip(runs[0], res):
            assert r0 == r1 and np.array_equal(g0, g1) and np.array_equal(p0, p1) and abs(l0 - l1) < 2e-6


def test_two_ranks_at_the_big_kernels_match_one_process_on_the_global_batch():
    """the same comparison with 16 x 256 tokens per rank — 4096 decoder (and encoder) rows, where the
    ping-pong NT kernel, the fused wi + GEGLU launch (from 4096 rows), the fused data-gradient + row kernels and the grouped
    weight-gradient launch (one per gradient bucket) all dispatch under two ranks; asserted through
    mrmt3_dispatch_counts.  Mel fed from the parent."""
    assert torch.cuda.is_available()
    res = _run_two_ranks(B=16, Ld=256, with_counts=True)
    (_, g0, p0, l0, c0, nb0), (_, g1, p1, l1, c1, nb1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    assert abs(l0 - l1) < 1e-6
    for c, nb in ((c0, nb0), (c1, nb1)):
        assert c["gemm_nt8"] >= 40, c                 # decoder forward + data-gradient products on the ping-pong kernel
        assert c["gemm_nt_geglu"] >= 16, c            # fused wi + GEGLU launches of both stacks
        assert c["tn_group"] >= nb - 1 >= 3, (c, nb)  # one grouped weight-gradient launch per bucket with gradients due
        assert c["gemm_nt_geglubwd"] >= 8 and c["gemm_nt_normbwd"] >= 8, c

    from mrmt3.trainer import Trainer
    dev = torch.device("cuda", 0)
    m = _model(dev)
    tr = Trainer(m, lr=1e-3, graph=False)
    a0, t0 = _batch(0, 16, 256)
    a1, t1 = _batch(1, 16, 256)
    loss = tr.train_step(torch.cat([a0, a1]).to(dev), torch.cat([t0, t1]).to(dev), audio=False)
    torch.cuda.synchronize()
    g = m.flat.G.cpu().numpy()
    rel = np.linalg.norm(g0 / 2 - g) / np.linalg.norm(g)
    assert rel < 2e-2, rel
    assert abs(loss.item() - l0) < 2e-3
    dp = np.abs(p0 - m.flat.P.cpu().numpy()).max()
    assert dp < 2.5e-3, dp


# ---- the segment-memory models under two ranks (BASELINE configs[2] and [4] are DDP configs of these models) ---------------

def _two_ranks_by_hand(variant, batches, steps, dev):
    """ONE process doing what two ranks do, one after the other: per step, each half's gradient from the same weights
    (the engine's own forward / fused lm_head + CE / backward, no collective ever enqueued), summed, then ONE AdamW step with
    grad_scale 1/2 — DDP's arithmetic (config/config.yaml:45) with nothing concurrent in it.  A rank's gradient is computed by the
    same kernels on the same shapes in both settings and a two-operand sum has one rounding, so the two-rank run must land on
    these bits EXACTLY: a bucket that left before its gradients were final (a deferred grouped weight gradient not joined,
    the side stream not waited for, the mid-backward segmem trigger raised too early) shows as a mismatch.
    The deferred weight gradients are joined where the ranks join them — at every bucket boundary — because the grouped
    launch's split of K depends on what is in the group: one group for the whole backward rounds differently in the last f32
    bit (1.5e-8 of 2.5e-2, profiles/r06_bucket_bits.txt), and AdamW's first steps turn such a bit into a full +-lr on the
    weights whose gradient is itself ~1e-8 (m / (sqrt(v) + eps) is a sign there)."""
    from mrmt3 import lib
    from mrmt3.trainer import Trainer
    m = _model(dev, variant)
    tr = Trainer(m, lr=1e-3, graph=False)
    eng, flat = tr.engine, tr.flat
    m.train()
    losses = []
    for _ in range(steps):
        gs = []
        for mel, lab, *rest in batches:
            mel, lab = mel.to(dev), lab.to(dev)
            prev = rest[0].to(dev) if rest else None
            eng.reset_deferred()
            eng._stream_ctr = 0
            dec, tape = eng.forward(mel, lab, prev, training=True, need_grad=True, want_logits=False)
            loss, dl = lib.lmhead_cross_entropy(dec, eng.W("lm_head"), lab.reshape(-1), want_grad=True,
                                                grad_dtype=torch.bfloat16)
            flat.G.zero_()

            def layer_done(prefix, i):
                if tr.buckets.triggered_by(prefix, i):
                    eng.join_wgrad()
            eng.backward(tape, dl, on_layer_done=layer_done)
            torch.cuda.synchronize()
            gs.append(flat.G.clone())
            losses.append(float(loss.item()))
        flat.G.copy_(gs[0] + gs[1])
        flat.adamw_step(tr.lr_dev, tr.step_dev, tr.betas, tr.eps, tr.wd, grad_scale=0.5)
    torch.cuda.synchronize()
    return flat.G.cpu().numpy(), flat.P.cpu().numpy(), 0.5 * (losses[-1] + losses[-2])


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "segmented_graph"])
@pytest.mark.parametrize("variant", ["with_prev", "v2"])
def test_two_ranks_of_the_segment_memory_models_equal_the_exchange_done_by_hand(variant, graph):
    """VERDICT r5 item 1.  MR-MT3's own model (memory from `targets_prev`, models/t5_segmem_v2_with_prev.py:118-128) and V2
    (memory from the previous batch ROW, models/t5_segmem_v2.py:126-137: row 0 of EACH rank gets the dummy memory, so the
    comparison is per half, not against the concatenated batch), two ranks on one GPU, three optimizer steps, eager and as
    one replayed graph per gradient bucket: the reduced gradient of step 3 and the weights after it are bit-equal to the
    exchange done by hand in one process (`_two_ranks_by_hand`), on both ranks.  The worker asserts the bucket layout:
    six buckets, the memory encoder's own, triggered mid-backward."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    batches = [_batch(r, 2, 128, prev=(variant == "with_prev")) for r in range(2)]
    steps = 5 if graph else 3                       # graph: 2 eager + capture + 2 replays
    want_g, want_p, want_loss = _two_ranks_by_hand(variant, batches, steps, dev)
    res = _run_two_ranks(steps=steps, graph=graph, variant=variant, batches=batches)
    (_, g0, p0, l0), (_, g1, p1, l1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    assert np.array_equal(g0, want_g), np.abs(g0 - want_g).max()
    assert np.array_equal(p0, want_p), np.abs(p0 - want_p).max()
    assert abs(l0 - want_loss) < 2e-6 and abs(l0 - l1) < 1e-6


@pytest.mark.parametrize("variant", ["with_prev", "v2"])
def test_two_ranks_of_the_segment_memory_models_at_the_big_kernels(variant):
    """The same at 16 x 256 tokens per rank: 4096 rows in every stack (the memory encoder's included), where the grouped
    weight-gradient launch is cut once per gradient bucket — the memory encoder's group must be complete when ITS bucket
    leaves between the decoder's and the encoder's backward."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    batches = [_batch(r, 16, 256, prev=(variant == "with_prev")) for r in range(2)]
    want_g, want_p, want_loss = _two_ranks_by_hand(variant, batches, 2, dev)
    res = _run_two_ranks(steps=2, variant=variant, batches=batches, with_counts=True)
    (_, g0, p0, l0, c0, nb0), (_, g1, p1, l1, c1, nb1) = res
    assert nb0 == nb1 == 6
    for c in (c0, c1):
        assert c["tn_group"] >= 2 * (nb0 - 1), c       # one grouped weight-gradient launch per bucket with gradients due, per step
        assert c["gemm_nt8"] >= 40 and c["gemm_nt_geglubwd"] >= 8 and c["gemm_nt_normbwd"] >= 8, c
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    assert np.array_equal(g0, want_g), np.abs(g0 - want_g).max()
    assert np.array_equal(p0, want_p), np.abs(p0 - want_p).max()
    assert abs(l0 - want_loss) < 2e-6


def test_two_ranks_of_mr_mt3_match_one_process_on_the_global_batch():
    """`targets_prev` makes every segment's memory its own, so two ranks on halves == one process on the concatenated batch
    (up to the bf16 products' rounding, which depends on the batch split) — the statement DDP makes for
    config_slakh_segmem_finetune.yaml."""
    assert torch.cuda.is_available()
    batches = [_batch(r, 2, 128, prev=True) for r in range(2)]
    res = _run_two_ranks(variant="with_prev", batches=batches)
    (_, g0, p0, l0), (_, g1, p1, l1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    from mrmt3.trainer import Trainer
    dev = torch.device("cuda", 0)
    m = _model(dev, "with_prev")
    tr = Trainer(m, lr=1e-3, graph=False)
    cat = [torch.cat([batches[0][i], batches[1][i]]).to(dev) for i in range(3)]
    loss = tr.train_step(cat[0], cat[1], cat[2], audio=False)
    torch.cuda.synchronize()
    g = m.flat.G.cpu().numpy()
    rel = np.linalg.norm(g0 / 2 - g) / np.linalg.norm(g)
    assert rel < 2e-2, rel
    seg = tr.buckets.buckets[5]
    gs, ge = g[seg["start"]:seg["end"]], g0[seg["start"]:seg["end"]] / 2
    assert np.linalg.norm(gs) > 0 and np.linalg.norm(ge - gs) / np.linalg.norm(gs) < 3e-2     # the memory encoder's own slice
    assert abs(loss.item() - l0) < 2e-3
    dp = np.abs(p0 - m.flat.P.cpu().numpy()).max()
    assert dp < 2.5e-3, dp


def _dropin_worker(rank, world, port, q, batch):
    """The reference's own training step shape (tasks/mt3_net.py: logits = model(...); CE; loss.backward()) under a
    multi-rank process group and WITHOUT mrmt3.trainer.Trainer."""
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
        from contrib import spectrograms as sp
        m = _model(dev).train()
        if rank == 1:
            with torch.no_grad():
                m.flat.P.mul_(1.5)                       # the first forward must start from rank 0's weights
        mel, lab = batch
        mel = mel.to(dev).float()
        out = m(inputs=mel, labels=lab.to(dev))
        loss = torch.nn.functional.cross_entropy(out.view(-1, out.shape[-1]), lab.to(dev).view(-1), ignore_index=-100)
        loss.backward()
        torch.cuda.synchronize()
        q.put((rank, m.flat.G.cpu().numpy(), m.flat.P.cpu().numpy(), float(loss.item())))
    finally:
        dist.destroy_process_group()


def test_dropin_backward_under_a_process_group_averages_the_gradients_itself():
    """VERDICT r1 missing #2: wrapped the way the reference's train.py does it (torch DDP via Lightning), nothing would
    reduce this module's gradients — its parameters never enter the autograd graph.  The module therefore exchanges
    them itself when a multi-rank group exists and no Trainer does: both ranks end up with the SAME gradient, the mean
    of the two local ones (= what DDP leaves in .grad), starting from rank 0's weights."""
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    batches = [_batch(r) for r in range(2)]
    torch.cuda.synchronize()
    procs = [ctx.Process(target=_dropin_worker, args=(r, 2, port, q, batches[r])) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, g0, p0, l0), (_, g1, p1, l1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)
    # single process, no group: the two local gradients, averaged by hand
    from contrib import spectrograms as sp
    dev = torch.device("cuda", 0)
    gs = []
    for r in range(2):
        m = _model(dev).train()
        mel, lab = batches[r]
        mel = mel.to(dev).float()
        out = m(inputs=mel, labels=lab.to(dev))
        torch.nn.functional.cross_entropy(out.view(-1, out.shape[-1]), lab.to(dev).view(-1), ignore_index=-100).backward()
        gs.append(m.flat.G.clone())
    want = ((gs[0] + gs[1]) * 0.5).cpu().numpy()
    assert np.allclose(g0, want, rtol=0, atol=1e-6 * np.abs(want).max() + 1e-12)
