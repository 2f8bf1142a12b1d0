"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI; gloo on CPU for tests).

Replaces what the reference gets implicitly from Lightning's `ddp_find_unused_parameters_false`
strategy (config/config.yaml:45): torch DDP's 25 MB reducer buckets over
per-parameter tensors.  Here gradients already live in ONE flat fp32 buffer in layer order, so a
bucket is just a contiguous slice: no flatten/unflatten copies, a handful of large all-reduces
(xGMI is point-to-point, 7 links per GPU — few, large messages), each launched asynchronously the
moment the backward pass has finished the layers it covers, so it overlaps the rest of backward.
The mean (1/world) is folded into the AdamW kernel's `grad_scale`; the logged loss (C4) is reduced by
the trainer with one more asynchronous collective that nothing on the host waits for.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def layer_ranges(flat) -> "List[Tuple[str, int, int]]":
    """(tag, start, end) element ranges of the flat buffer, tag = 'encoder.3', 'decoder.7', 'head', ..."""
    out = []
    cur_tag, start, end = None, 0, 0
    for key, off in flat.offsets.items():
        n = 1
        for s in flat.shapes[key]:
            n *= s
        parts = key.split(".")
        if parts[0] == "decoder" and parts[1] == "block" and parts[5] == "EncDecAttention" and parts[6] in ("k", "v"):
            tag = "decoder.ckv"          # the cross k|v projections of all layers sit together (params.py)
        elif parts[0] in ("encoder", "decoder", "segmem_encoder") and parts[1] == "block":
            tag = f"{parts[0]}.{parts[2]}"
        elif parts[0] in ("encoder", "decoder", "segmem_encoder"):
            tag = f"{parts[0]}.final"
        else:
            tag = parts[0]
        if tag != cur_tag:
            if cur_tag is not None:
                out.append((cur_tag, start, end))
            cur_tag, start = tag, off
        end = off + n
    out.append((cur_tag, start, end))
    return out


class _Widen:
    """A pending compressed bucket: wait() = wait for the collective, then copy the reduced values back as f32."""

    def __init__(self, work, low, grad):
        self.work, self.low, self.grad = work, low, grad

    def wait(self):
        self.work.wait()
        self.grad.copy_(self.low)


class _StreamWork:
    """A collective enqueued on the launch stream through the C ABI (MRMT3_DDP_NATIVE=1): wait() = the current stream
    waits for it (what torch's Work.wait() does for an RCCL collective: a stream-level wait, nothing on the host)."""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)


class GradBuckets:
    """Partition of the flat gradient buffer into buckets ordered by when backward completes them."""

    def __init__(self, flat, n_layers_enc: int, n_layers_dec: int, has_segmem: bool, layers_per_bucket: int = 2,
                 group=None, exchange_dtype=None):
        self.flat, self.group = flat, group
        # exchange_dtype=torch.bfloat16: each bucket is rounded to bf16, all-reduced in bf16 and widened back — half
        # the bytes on the xGMI links (91.8 instead of 183.6 MB per step for MT3Net); the local f32 gradient is replaced
        # by the reduced bf16 one, i.e. every rank still ends up with identical gradients.  Default: f32, like torch DDP.
        self.exchange_dtype = exchange_dtype
        self._staging = None
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        rng = {t: (a, b) for t, a, b in layer_ranges(flat)}
        self.ranges = rng
        # completion order of backward: head -> decoder L-1..0 -> (segmem) -> encoder L-1..0 -> inputs
        self.buckets: "List[dict]" = []

        def add(tags, trigger):
            a = min(rng[t][0] for t in tags)
            b = max(rng[t][1] for t in tags)
            assert sum(rng[t][1] - rng[t][0] for t in tags) == b - a, "bucket must be contiguous"
            self.buckets.append(dict(tags=tags, start=a, end=b, trigger=trigger))

        for hi in range(n_layers_dec - 1, -1, -layers_per_bucket):
            lo = max(hi - layers_per_bucket + 1, 0)
            tags = [f"decoder.{i}" for i in range(lo, hi + 1)]
            if hi == n_layers_dec - 1:
                tags += ["decoder.final", "lm_head"]
            if lo == 0 and "decoder.ckv" in rng:
                tags = ["decoder.ckv"] + tags          # complete when the last (lowest) decoder layer is done
            add(tags, ("decoder", lo))
        # The LAST bucket's all-reduce is the one nothing overlaps (backward has ended): it holds only the encoder's lowest
        # `last_layers` layers + the embedding tables (MRMT3_DDP_LAST_BUCKET_LAYERS, default 1: 16 MB instead of the 42 MB of a
        # full four-layer bucket; one more boundary, a shorter exposed tail — profiles/r05_overlap_emulation.txt).
        last_layers = max(1, min(layers_per_bucket, int(os.environ.get("MRMT3_DDP_LAST_BUCKET_LAYERS", "1"))))
        enc_cuts = []                       # (hi, lo) encoder layer ranges, in completion order
        hi = n_layers_enc - 1
        while hi >= last_layers:
            lo = max(hi - layers_per_bucket + 1, last_layers)
            enc_cuts.append((hi, lo))
            hi = lo - 1
        if hi >= 0:
            enc_cuts.append((hi, 0))
        for hi, lo in enc_cuts:
            tags = [f"encoder.{i}" for i in range(lo, hi + 1)]
            if hi == n_layers_enc - 1:
                tags += ["encoder.final"]
            if lo == 0:
                # the embedding tables finish last (decoder_embed after the segmem path, proj at the very end)
                tags = ["proj", "decoder_embed_tokens"] + tags
                add(tags, ("end", 0))
            else:
                add(tags, ("encoder", lo))
        if has_segmem:
            # the memory encoder's backward runs between the decoder's and the encoder's (Engine.backward): its gradients are
            # final there, and their all-reduce hides under the encoder's backward instead of joining the exposed tail
            seg = [t for t in rng if t.startswith("segmem")]
            add(seg, ("segmem", 0))
        covered = sorted((b["start"], b["end"]) for b in self.buckets)
        pos = 0
        for a, b in covered:
            assert a == pos, "buckets must tile the flat buffer"
            pos = b
        assert pos == flat.numel
        self._works = []
        self._fired = set()
        # called right before a bucket's all-reduce is enqueued (the engine sums its queued norm-weight gradients here)
        self.before_fire = None
        # streams other than the current one that also write gradients (weight gradients are produced on a second
        # stream, see Engine.wgrad).  The collective is enqueued from a launch stream that waits for the current
        # stream AND these — the backward pass itself never stops to wait for its own side stream at a bucket boundary.
        self.producer_streams = None
        self._launch = None
        # single-GPU check of the collective path (profiles/tools/nccl_one_rank_check.py): run the all-reduces at world 1
        self.force = os.environ.get("MRMT3_DDP_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()
        # MRMT3_DDP_NATIVE=1: the buckets of GPU gradients go through the C ABI (mrmt3_allreduce on an RCCL communicator of the
        # library's own, csrc/comm.hip) instead of torch.distributed; torch.distributed only carries the 128-byte id once
        self.native = os.environ.get("MRMT3_DDP_NATIVE") == "1"
        self._comm = None

    def reset(self):
        self._works, self._fired = [], set()

    @property
    def active(self) -> bool:
        """Will firing a bucket enqueue a collective?"""
        return self.world > 1 or self.force

    def triggered_by(self, prefix, i):
        """Indices of the buckets that the completion of `prefix` layer i finishes."""
        return [idx for idx, b in enumerate(self.buckets) if b["trigger"] == (prefix, i)]

    def fire(self, idx):
        self._fire(idx)

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []

    def _native_comm(self):
        if self._comm is None:
            from . import lib
            rank = dist.get_rank(self.group)
            box = [lib.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0 if self.group is None else dist.get_global_rank(self.group, 0), group=self.group)
            self._comm = lib.Comm(box[0], rank, self.world)
        return self._comm

    def _all_reduce(self, t, stream=None):
        """Enqueue the in-place sum of `t` over the ranks (on `stream`, default the current one); returns its Work."""
        if self.native and t.is_cuda:
            s = stream
            if s is None:       # never on the compute stream itself: the collective overlaps what backward enqueues next
                s = self._launch_stream(t.device)
                s.wait_stream(torch.cuda.current_stream(t.device))
            self._native_comm().allreduce(t, stream=s)
            ev = torch.cuda.Event()
            ev.record(s)
            return _StreamWork(ev, t.device)
        if stream is None:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        with torch.cuda.stream(stream):
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def comm(self):
        """The library's own RCCL communicator (created on first use: every rank must get here together)."""
        return self._native_comm()

    def collective_stream(self, device):
        """The stream the buckets' collectives run on (never the compute stream)."""
        return self._launch_stream(device)

    def use_collective_stream(self, stream):
        """Run the collectives on `stream` from now on (the trainer picks one that shares no hardware queue with compute)."""
        self._launch = stream

    def _launch_stream(self, device):
        if self._launch is None or self._launch.device != device:
            # Normal priority, like every round so far (MRMT3_DDP_STREAM_PRIO=-1: high).  Two things were learnt about this
            # stream in round 5 (profiles/r05_two_graph_probe.txt, r05_collectives_ab.txt): HIP deals the streams of one
            # priority over a few hardware queues, so it MAY share the compute stream's queue — then the collectives queue
            # between the backward kernels and overlap nothing, which is why the trainer tests and, if need be, replaces this
            # stream before the first exchange (Trainer._pick_collective_stream); and a kernel resident on a second queue
            # costs the compute graph ~9 us per dependent launch, ~20 % more when that queue is a high-priority one.
            self._launch = torch.cuda.Stream(device=device, priority=int(os.environ.get("MRMT3_DDP_STREAM_PRIO", "0")))
        return self._launch

    def close(self):
        """Release the library's communicator (native path); the buckets stay usable — the next exchange makes a new one.
        (No collective is ever captured into a graph, so no graph can outlive the communicator whose kernels it holds;
        owners call Trainer.close(), which drops the step's graphs first all the same.)"""
        if self._comm is not None:
            self._comm.close()
            self._comm = None

    def _fire(self, idx):
        if idx in self._fired:
            return
        self._fired.add(idx)
        if self.world == 1 and not self.force:
            return
        b = self.buckets[idx]
        if self.before_fire is not None:
            self.before_fire()
        grad = self.flat.G[b["start"]:b["end"]]
        if self.exchange_dtype is not None and self.exchange_dtype != grad.dtype:
            return self._fire_compressed(b, grad)
        extra = [s for s in (self.producer_streams() if self.producer_streams is not None else []) if s is not None]
        if grad.is_cuda and extra:
            launch = self._launch_stream(grad.device)
            launch.wait_stream(torch.cuda.current_stream(grad.device))
            for s in extra:
                launch.wait_stream(s)
            self._works.append(self._all_reduce(grad, launch))
        else:
            self._works.append(self._all_reduce(grad))

    def _fire_compressed(self, b, grad):
        """Round the bucket to the exchange dtype, all-reduce that copy, widen it back into the f32 gradient when the
        collective has finished (`wait()`).  The copies run on the current stream, behind the bucket's producers."""
        if grad.is_cuda:
            for s in (self.producer_streams() if self.producer_streams is not None else []):
                if s is not None:
                    torch.cuda.current_stream(grad.device).wait_stream(s)
        if self._staging is None or self._staging.device != grad.device:
            self._staging = torch.empty(self.flat.numel, dtype=self.exchange_dtype, device=grad.device)
        low = self._staging[b["start"]:b["end"]]
        low.copy_(grad)
        self._works.append(_Widen(self._all_reduce(low), low, grad))

    def on_layer_done(self, prefix, i):
        """Engine callback: every gradient of `prefix` layer i (and above) is final."""
        for idx, b in enumerate(self.buckets):
            if b["trigger"] == (prefix, i):
                self._fire(idx)

    def finish(self):
        """Backward is complete: launch whatever is left and wait for every collective."""
        for idx in range(len(self.buckets)):
            self._fire(idx)
        for w in self._works:
            w.wait()
        self._works = []
