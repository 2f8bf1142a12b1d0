"""N>1 path on CPU: world_size-2 gloo run of the gradient-bucket exchange (mrmt3.ddp.GradBuckets),
plus structural checks of the bucket partition.  The HIP kernels are not involved: the exchange is
pure torch.distributed plumbing over slices of the flat gradient buffer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mrmt3.ddp import GradBuckets, layer_ranges
from mrmt3.params import FlatParams
from mrmt3.synthetic import T5_SMALL

SMALL = dict(T5_SMALL, num_layers=4, num_decoder_layers=4, d_ff=256, vocab_size=256)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("seg", [0, 1])
def test_buckets_tile_the_flat_buffer(seg):
    flat = FlatParams(T5_SMALL, seg)
    gb = GradBuckets(flat, 8, 8, bool(seg))
    spans = sorted((b["start"], b["end"]) for b in gb.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == flat.numel
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert flat.numel == (48519680 if seg else 45896704)
    # first bucket to fire holds lm_head + the last decoder layers; ~184 MB fp32 in a handful of messages
    assert "lm_head" in gb.buckets[0]["tags"] and len(gb.buckets) <= 10
    tags = [t for t, _, _ in layer_ranges(flat)]
    assert tags[:3] == ["proj", "decoder_embed_tokens", "encoder.0"]
    # round 5 layout (chosen under emulated collectives, profiles/r05_overlap_emulation.txt): the LAST bucket — the one all-reduce
    # nothing overlaps — holds only the encoder's lowest layer and the embedding tables (14 MB, not 42); the memory encoder's
    # bucket leaves after ITS backward, which runs between the decoder's and the encoder's (Engine.backward)
    gb = GradBuckets(flat, 8, 8, bool(seg), 4)                          # the trainer's 4 layers per bucket
    last = [b for b in gb.buckets if b["trigger"] == ("end", 0)]
    assert len(last) == 1 and sorted(last[0]["tags"]) == ["decoder_embed_tokens", "encoder.0", "proj"]
    assert (last[0]["end"] - last[0]["start"]) * 4 < 16e6
    assert [b["trigger"] for b in gb.buckets if any(t.startswith("encoder") for t in b["tags"])] == [("encoder", 4), ("encoder", 1), ("end", 0)]
    if seg:
        mem = [b for b in gb.buckets if any(t.startswith("segmem") for t in b["tags"])]
        assert len(mem) == 1 and mem[0]["trigger"] == ("segmem", 0) and all(t.startswith("segmem") for t in mem[0]["tags"])


def _worker(rank, world, port, q, exchange=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat = FlatParams(SMALL, 1)
        flat.ensure_grads()
        g = torch.Generator().manual_seed(100 + rank)
        flat.G.copy_(torch.randn(flat.numel, generator=g))
        mine = flat.G.clone()
        gb = GradBuckets(flat, 4, 4, True, exchange_dtype=exchange)
        gb.reset()
        # backward order: decoder layers high -> low, then encoder, then the rest at finish()
        for i in reversed(range(4)):
            gb.on_layer_done("decoder", i)
        gb.on_layer_done("segmem", 0)            # (the memory encoder's backward sits between the two stacks')
        for i in reversed(range(4)):
            gb.on_layer_done("encoder", i)
        gb.finish()
        other = torch.randn(flat.numel, generator=torch.Generator().manual_seed(100 + (1 - rank)))
        if exchange is None:
            ok = torch.allclose(flat.G, mine + other, atol=1e-6)
        else:
            # bf16 exchange: each rank's bucket is rounded to bf16, the sum is formed and rounded in bf16
            want = (mine.to(exchange).float() + other.to(exchange).float())
            ok = bool(((flat.G - want).abs() <= want.abs() * 2.0 ** -8 + 1e-30).all()) and \
                torch.equal(flat.G, flat.G.to(exchange).float())
            ok = ok and ((flat.G - (mine + other)).norm() / (mine + other).norm()).item() < 6e-3
        # every rank ends with identical reduced gradients
        chk = flat.G.double().sum().reshape(1)
        lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(lst, chk)
        q.put((rank, bool(ok), bool(torch.equal(lst[0], lst[1]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", [None, torch.bfloat16])
def test_two_rank_gloo_bucketed_allreduce(exchange):
    """f32 exchange: the exact sum.  bf16 exchange (half the bytes on the links): the reduced gradient is the bf16 sum
    of the bf16-rounded buckets — within bf16 rounding of the f32 sum, identical on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok and same for _, ok, same in res), res
