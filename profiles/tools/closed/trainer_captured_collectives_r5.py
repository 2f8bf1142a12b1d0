"""Training step of the MI355X path: audio -> log-mel -> forward -> fused CE -> hand-written backward
(overlapped with the RCCL gradient exchange) -> one-launch AdamW.

This is the build's counterpart of what `train.py:43-103` gets from `pl.Trainer.fit` +
`MT3Net*.training_step` + `configure_optimizers` (tasks/mt3_net.py:27-68): same loss, same
optimizer semantics (torch.optim.AdamW defaults, cosine-warmup LambdaLR stepped per batch), one
process per GPU.  Nothing on the host waits for the device inside a step: the learning rate, the
step counter and the loss stay in device memory.

Host out of the step.  After two eager steps of a given input shape the whole step — ≈700 kernel launches on two
streams with their event fork/joins — is captured into hipGraphs and replayed: the host then enqueues a handful of
graph launches per step instead of ≈600 ctypes calls (19.6 of 28 ms per step in round 1).  What makes the replay equal
to the eager step bit for bit:
  * dropout masks are keyed by (seed, site, DEVICE step counter): the site ids restart at 0 every step and the
    kernels read `step_dev`, which AdamW increments, so frozen by-value arguments still give new masks each step;
  * the learning rate is written to `lr_dev` (device) before the replay, outside the graph;
  * inputs are copied into static buffers the captured kernels read.
With more than one rank the step is cut into one graph per gradient bucket: the RCCL all-reduce of a finished bucket
is enqueued EAGERLY between two replays on the launch stream, so it still overlaps the rest of backward and no
collective is ever captured.  That is the default, because it is the form that needs nothing from RCCL but a plain call.

Round 5 built the two captured forms the round-4 review asked for, as switches (MRMT3_DDP_GRAPH), and measured them at one
rank with forced collectives (profiles/r05_collectives_ab.txt; neither can be measured at N > 1 here):
  "inline"  ONE graph, every bucket's all-reduce (mrmt3_allreduce on the library's own RCCL communicator) a node of the compute
            chain: no segment per bucket (+0.13-0.25 ms over the plain step instead of +0.27-0.35), but the collective is then
            exposed instead of hidden under the rest of backward;
  "1"       TWO graphs replayed side by side on two streams — the whole compute step as one chain, the buckets' all-reduces
            as a second chain on the collective stream — that talk through counting flags in device memory
            (mrmt3_flag_signal / mrmt3_flag_wait, csrc/comm.hip): "bucket i complete" from the compute chain, "all reduced"
            back before AdamW.  Not one graph with a side branch: ROCm 7.2 replays forked graphs serially and slowly (DESIGN
            §3).  Correct (bit-equal to the eager bucketed step) and 3.2-3.8 ms SLOWER per step: while a second hardware
            queue holds a resident kernel, every dependent launch of the compute graph costs ~9 us more.
So the default stays the segmented form.  If the capture of a collective fails, or no stream can be found that runs side by
side with the compute stream (HIP shares a few hardware queues among the streams of a priority:
profiles/r05_two_graph_probe.txt), the trainer falls back to the segmented form.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import lib
from .ddp import GradBuckets


class _CapturedStep:
    """One input signature's captured step: graph segments (each followed by the gradient buckets to send), the tail
    graph (AdamW) and the static tensors the graphs read and write."""

    def __init__(self):
        self.segments, self.tail = [], None
        self.comm = None          # two-graph form: the chain of all-reduces, replayed on the collective stream
        self.inputs = self.labels = self.prev = self.loss = None


class Trainer:
    def __init__(self, model, lr: float = 2e-4, lr_lambda=None, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, weighted_loss: bool = False, layers_per_bucket: int = 4,
                 graph: bool = None, grad_exchange_dtype=None):
        self.model, self.flat, self.engine = model, model.flat, model.engine
        assert model.device.type == "cuda", "the trainer drives the HIP kernels: move the model to the GPU first"
        self.base_lr, self.lr_lambda = lr, lr_lambda
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.weighted = weighted_loss
        dev = model.device
        self.lr_dev = torch.full((1,), lr, device=dev, dtype=torch.float32)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.host_step = 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        cfg = model.cfg
        if grad_exchange_dtype is None and os.environ.get("MRMT3_GRAD_EXCHANGE", "f32") == "bf16":
            grad_exchange_dtype = torch.bfloat16
        # 4 layers per bucket, the last bucket cut down to the encoder's lowest layer + the embedding tables (ddp.py): 5 buckets of
        # 47 / 57 / 38 / 28 / 14 MB for MT3Net (6 with segment memory).  A bucket boundary costs 0.07 ms
        # of step time (a graph segment of its own + the grouped weight-gradient launch split there: 16 / 8 / 4 / 2 buckets =
        # 25.26 / 24.70 / 24.40 / 24.23 ms at one rank with forced collectives, plain step 24.17,
        # profiles/r04_bucket_boundary_cost.txt); the LAST bucket's all-reduce is the one nothing overlaps, so fewer, larger
        # buckets stop paying once that tail outgrows the boundaries saved (2 buckets: the whole encoder, 80 MB, at the end).
        layers_per_bucket = max(1, int(os.environ.get("MRMT3_DDP_LAYERS_PER_BUCKET", layers_per_bucket)))
        self.buckets = GradBuckets(self.flat, cfg["num_layers"], cfg["num_decoder_layers"],
                                   model.segmem_num_layers > 0, layers_per_bucket, exchange_dtype=grad_exchange_dtype)
        self.buckets.before_fire = model.engine.join_wgrad      # norm-weight partials and split-K slabs are summed here
        self.buckets.producer_streams = lambda: [model.engine._side]
        self.flat.ensure_grads()
        self.flat.ensure_adam()
        if self.world > 1:   # C2: identical replicas
            dist.broadcast(self.flat.P, src=0)
        self.last_loss = None
        # every dropout mask of a step is salted in-kernel by the device step counter (see module docstring)
        self.engine.step_dev = self.step_dev
        self.use_graph = (os.environ.get("MRMT3_TRAIN_GRAPH", "1") != "0") if graph is None else bool(graph)
        # "" (default): one graph per gradient bucket, collectives eager between them; "1": two graphs side by side, the
        # collectives captured (needs the f32 exchange and the library's own communicator); "inline": one graph, collectives
        # in the compute chain (A/B only).  See the module docstring.
        self.ddp_graph = os.environ.get("MRMT3_DDP_GRAPH", "")
        if self.ddp_graph in ("0", "off"):
            self.ddp_graph = ""
        if self.ddp_graph and self.buckets.exchange_dtype is not None:
            self.ddp_graph = ""                              # the compressed exchange stages copies around the collective
        if self.ddp_graph and self.buckets.active:
            self.buckets.native = True                       # the eager warm-up steps use the communicator the capture will
        self._hand = None                                    # flags of the two-graph form (device int32): see _handoffs()
        self._collective_stream_checked = False
        self.graph_warmup = 2            # eager steps per input signature before capture (tables, workspaces)
        self._graphs = {}                # signature -> _CapturedStep
        self._eager_seen = {}
        self._cap_stream = None

    def mel_from_audio(self, audio):
        """[B, n_samples] f32 device audio -> [B, frames, 512] mel in the compute dtype."""
        from contrib import spectrograms as sp
        return sp.logmel_segments(audio, out_bf16=(self.engine.dt == torch.bfloat16))

    # ---- one step's device work (identical in eager mode, under capture and — by replay — afterwards) ------------
    def _step_body(self, inputs, labels, targets_prev, audio, cut=None, before_optimizer=None):
        """Enqueues one optimizer step.  `cut(bucket_indices)` is called where a gradient bucket is complete (only
        when collectives will run); under capture it closes the current graph segment (or, with the collectives
        captured, signals the bucket to the collective graph).  `before_optimizer()` runs right before AdamW."""
        eng, flat = self.engine, self.flat
        eng.reset_deferred()                                 # nothing of an aborted capture / failed step leaks into this one
        eng._stream_ctr = 0                                  # dropout site ids are per-step (step_dev salts them)
        mel = self.mel_from_audio(inputs) if audio else inputs
        if eng.dt == torch.bfloat16:
            dec, tape = eng.forward(mel, labels, targets_prev, training=True, need_grad=True, want_logits=False)
            # lm_head + CE over row chunks: the f32 logits exist one chunk at a time in a cache-sized workspace
            loss, dl = lib.lmhead_cross_entropy(dec, eng.W("lm_head"), labels.reshape(-1), want_grad=True,
                                                grad_dtype=torch.bfloat16, weighted=self.weighted)
        else:                                                # fp32 engine (`precision: 32`): exact-f32 lm_head, then CE
            logits, tape = eng.forward(mel, labels, targets_prev, training=True, need_grad=True)
            loss, dl = lib.cross_entropy(logits.view(-1, logits.shape[-1]), labels.reshape(-1), want_grad=True,
                                         grad_dtype=torch.float32, weighted=self.weighted)
        flat.G.zero_()
        self.buckets.reset()
        if cut is None:
            eng.backward(tape, dl, on_layer_done=self.buckets.on_layer_done)
            self.buckets.finish()
        else:
            sent = set()

            def layer_done(prefix, i):
                idx = [j for j in self.buckets.triggered_by(prefix, i) if j not in sent]
                if idx:
                    sent.update(idx)
                    eng.join_wgrad()                         # a capture must end with its forked stream joined
                    cut(idx)
            active = self.buckets.active
            eng.backward(tape, dl, on_layer_done=layer_done if active else None)     # ends with join_wgrad()
            cut([j for j in range(len(self.buckets.buckets)) if j not in sent] if active else [])
        if before_optimizer is not None:
            before_optimizer()
        flat.adamw_step(self.lr_dev, self.step_dev, self.betas, self.eps, self.wd, grad_scale=1.0 / self.world)
        return loss

    def train_step(self, inputs, labels, targets_prev=None, audio: bool = False):
        """One optimizer step.  `inputs` is mel [B,Le,512] or, with audio=True, raw audio [B,n].
        Returns the (device, un-synchronised) mean loss of this rank."""
        m, eng = self.model, self.engine
        m.train()
        if self.buckets.active and not self._collective_stream_checked:
            # once, before the first exchange: the collectives only overlap the rest of backward if their stream sits on
            # another hardware queue than the compute stream's (profiles/r05_two_graph_probe.txt)
            self._collective_stream_checked = True
            if inputs.is_cuda and not self._pick_collective_stream(torch.cuda.current_stream(), inputs.device):
                import warnings
                warnings.warn("no stream was found that runs side by side with the compute stream: the gradient all-reduces "
                              "will queue behind the backward kernels instead of overlapping them")
        if self.lr_lambda is not None:
            self.lr_dev.fill_(self.base_lr * self.lr_lambda(self.host_step))
        if targets_prev is not None and eng.variant == "segmem_v2_with_prev":
            # in place on the caller's tensor, like the reference (t5_segmem_v2_with_prev.py:119)
            targets_prev.masked_fill_(targets_prev == -100, m.cfg["pad_token_id"])
        if self.use_graph:
            loss = self._graph_step(inputs, labels, targets_prev, audio)
        else:
            loss = self._step_body(inputs, labels, targets_prev, audio)
        self.host_step += 1
        if self.world > 1:   # C4: logged loss, reduced without blocking the host
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, async_op=True).wait()   # stream-level wait only
            loss /= self.world
        self.last_loss = loss
        return loss

    # ---- hipGraph capture / replay of the step ---------------------------------------------------------------
    def _graph_step(self, inputs, labels, targets_prev, audio):
        sig = (bool(audio), tuple(inputs.shape), inputs.dtype, tuple(labels.shape),
               None if targets_prev is None else tuple(targets_prev.shape))
        cap = self._graphs.get(sig)
        if cap is None:
            seen = self._eager_seen.get(sig, 0)
            if seen < self.graph_warmup:
                self._eager_seen[sig] = seen + 1
                return self._step_body(inputs, labels, targets_prev, audio)
            try:
                try:
                    cap = self._capture(sig, inputs, labels, targets_prev, audio)
                except Exception as e:     # noqa: BLE001
                    if not (self.ddp_graph and self.buckets.active):
                        raise
                    # the collectives would not capture: the segmented form (collectives eager between the segments)
                    import warnings
                    warnings.warn("capturing the gradient all-reduces failed (%s: %s); falling back to one graph per "
                                  "bucket with eager collectives" % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
                    self.ddp_graph = ""
                    self.engine.reset_deferred()
                    self.engine._stream_ctr = 0
                    torch.cuda.synchronize()
                    cap = self._capture(sig, inputs, labels, targets_prev, audio)
            except Exception as e:     # noqa: BLE001 — whatever a capture trips over, the eager step is still correct
                # (nothing executed during the failed capture: the step below is the first to run; the launches the
                # aborted capture had deferred are dropped — _step_body starts with Engine.reset_deferred())
                import warnings
                warnings.warn("hipGraph capture of the training step failed (%s: %s); continuing with eager launches"
                              % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
                self.use_graph = False
                self.engine._stream_ctr = 0
                torch.cuda.synchronize()
                return self._step_body(inputs, labels, targets_prev, audio)
        self.engine.prepare(True)          # weights written through torch since the last step? rebuild the shadows
        cap.inputs.copy_(inputs, non_blocking=True)
        cap.labels.copy_(labels, non_blocking=True)
        if cap.prev is not None:
            cap.prev.copy_(targets_prev, non_blocking=True)
        self.buckets.reset()
        if cap.comm is not None:           # two graphs side by side; they order themselves through the hand-off flags
            ks = self.buckets.collective_stream(cap.inputs.device)
            with torch.cuda.stream(ks):
                cap.comm.replay()
        for graph, fire in cap.segments:
            graph.replay()
            for idx in fire:
                self.buckets.fire(idx)
        self.buckets.wait()
        cap.tail.replay()
        return cap.loss.clone()            # the graph's own loss scalar is overwritten by the next replay

    # ---- the collectives captured: hand-off flags, the second graph ------------------------------------------------
    def _handoffs(self, device):
        """Device words of the two-graph form: flags[j] counts completions of bucket j (compute graph), flags[n] counts
        "every bucket reduced" (collective graph); seen[] are the waiting sides' own counters; err is raised by a wait that
        timed out (check_exchange())."""
        if self._hand is None:
            n = len(self.buckets.buckets) + 1
            self._hand = dict(flags=torch.zeros(n, dtype=torch.int32, device=device),
                              seen=torch.zeros(n, dtype=torch.int32, device=device),
                              err=torch.zeros(1, dtype=torch.int32, device=device),
                              timeout_ms=int(os.environ.get("MRMT3_DDP_GRAPH_TIMEOUT_MS", "20000")))
        return self._hand

    def _side_by_side(self, compute_stream, collective_stream) -> bool:
        """Do kernels of the two streams run side by side?  A spinning wait on the collective stream, then its signal on
        the compute stream: if the wait times out, both streams feed ONE hardware queue (HIP shares a few queues among
        the streams of a priority) and the two-graph form must not be used.  Eager, once per capture, 0.1 ms when fine."""
        dev = self.flat.G.device
        w = torch.zeros(3, dtype=torch.int32, device=dev)            # flag, seen, err
        torch.cuda.synchronize()
        lib.flag_wait(w[0:1], w[1:2], w[2:3], 250, stream=collective_stream)
        lib.flag_signal(w[0:1], stream=compute_stream)
        torch.cuda.synchronize()
        return int(w[2].item()) == 0

    def _pick_collective_stream(self, compute_stream, device) -> bool:
        """A collective stream on ANOTHER hardware queue than the compute stream's: on a shared queue an eager all-reduce
        simply queues between the backward kernels (no overlap), and the two-graph form's spinning hand-off waits would
        block the kernels they wait for.  HIP deals the streams of one priority over a few queues, so: test the stream the
        buckets already use, then up to eight fresh ones (MRMT3_DDP_STREAM_PRIO: their priority, default normal — a
        resident kernel on a HIGH-priority queue slows the compute graph's launches more, profiles/r05_collectives_ab.txt)
        and keep the first that passes.  False: none did."""
        prio = int(os.environ.get("MRMT3_DDP_STREAM_PRIO", "0"))
        first = self.buckets.collective_stream(device)
        cands = ([first] if first.priority == prio else []) + [None] * 8
        self._stream_candidates = []
        for c in cands:
            s = c if c is not None else torch.cuda.Stream(device=device, priority=prio)
            self._stream_candidates.append(s)
            if s.cuda_stream != compute_stream.cuda_stream and self._side_by_side(compute_stream, s):
                self.buckets.use_collective_stream(s)
                return True
        return False

    def check_exchange(self):
        """Raises if a hand-off between the compute graph and the collective graph ever timed out (host sync: call it where
        the host waits anyway — end of an epoch, a checkpoint, the end of a benchmark)."""
        if self._hand is not None and int(self._hand["err"].item()) != 0:
            raise RuntimeError("data-parallel step: a graph hand-off timed out (a gradient bucket was never signalled or "
                               "never reduced); the gradients of that step are not the all-reduced ones")

    def _capture(self, sig, inputs, labels, targets_prev, audio):
        """Record the step once (nothing executes during capture); `train_step` then replays it, this step included."""
        eng = self.engine
        cur = torch.cuda.current_stream()
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
        cs = self._cap_stream
        torch.cuda.synchronize()           # nothing of the eager steps (collectives included) is in flight during capture
        eng.prepare(True)
        cap = _CapturedStep()
        cap.inputs, cap.labels = inputs.clone(), labels.clone()
        cap.prev = None if targets_prev is None else targets_prev.clone()
        cs.wait_stream(cur)
        pool = None
        state = {"g": None}

        def begin():
            # thread-local capture mode: the process group's watchdog thread polls the events of earlier collectives
            # (hipEventQuery) whenever it likes; under the default global mode that call is illegal while ANY thread
            # captures and the watchdog takes the process down (seen with RCCL at world size 1, forced collectives)
            g = torch.cuda.CUDAGraph()
            if pool is None:
                g.capture_begin(capture_error_mode="thread_local")
            else:
                g.capture_begin(pool=pool, capture_error_mode="thread_local")
            state["g"] = g

        def cut(fire):
            nonlocal pool
            g = state["g"]
            g.capture_end()
            if pool is None:
                pool = g.pool()
            cap.segments.append((g, list(fire)))
            begin()

        overlap_was = eng.overlap_wgrad
        if os.environ.get("MRMT3_GRAPH_LINEAR", "1") == "1":
            eng.overlap_wgrad = False          # one chain of nodes, no fork/join edges in the graph
        mode = self.ddp_graph if self.buckets.active else ""
        before_opt, order = None, []
        if mode and mode != "inline" and not self._pick_collective_stream(cur, cap.inputs.device):
            raise RuntimeError("no stream was found that runs side by side with the compute stream (shared hardware queues): "
                               "two graphs with spinning hand-offs between them would block each other")
        if mode:
            comm = self.buckets.comm()         # created (and used by the eager steps) before anything captures
            G = self.flat.G
            if mode == "inline":
                def cut(fire):                 # noqa: F811 — the collective as a node of the compute chain itself
                    for j in fire:
                        b = self.buckets.buckets[j]
                        comm.allreduce(G[b["start"]:b["end"]], stream=cs)
            else:
                hand = self._handoffs(cap.inputs.device)
                n_b = len(self.buckets.buckets)

                def cut(fire):                 # noqa: F811 — "bucket j is complete" to the collective graph
                    for j in fire:
                        lib.flag_signal(hand["flags"][j:j + 1], stream=cs)
                        order.append(j)

                def before_opt():
                    lib.flag_wait(hand["flags"][n_b:], hand["seen"][n_b:], hand["err"], hand["timeout_ms"], stream=cs)
        with torch.cuda.stream(cs):
            begin()
            try:
                cap.loss = self._step_body(cap.inputs, cap.labels, cap.prev, audio, cut=cut, before_optimizer=before_opt)
                state["g"].capture_end()
            except Exception:
                try:
                    state["g"].capture_end()
                except Exception:
                    pass
                raise
            finally:
                eng.overlap_wgrad = overlap_was
            cap.tail = state["g"]
        if mode and mode != "inline":
            # the second graph: for every bucket in the order backward completes them — wait for its signal, all-reduce it;
            # then "all reduced" back to the compute chain
            ks = self.buckets.collective_stream(cap.inputs.device)
            ks.wait_stream(cs)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.stream(ks):
                g2.capture_begin(pool=pool if pool is not None else cap.tail.pool(), capture_error_mode="thread_local")
                try:
                    for j in order:
                        b = self.buckets.buckets[j]
                        lib.flag_wait(hand["flags"][j:j + 1], hand["seen"][j:j + 1], hand["err"], hand["timeout_ms"], stream=ks)
                        comm.allreduce(G[b["start"]:b["end"]], stream=ks)
                    lib.flag_signal(hand["flags"][n_b:], stream=ks)
                finally:
                    g2.capture_end()
            assert sorted(order) == list(range(n_b)), order
            cap.comm = g2
            cs.wait_stream(ks)
        cur.wait_stream(cs)
        self._graphs[sig] = cap
        return cap

    @property
    def graph_captured(self) -> bool:
        return bool(self._graphs)

    # ---- checkpoint / resume (Lightning `.ckpt` layout, see mrmt3.checkpoint) -------------------------------
    def save_checkpoint(self, path: str, epoch: int = 0):
        """Write weights + AdamW moments + step in the layout the reference's ModelCheckpoint produces, so
        either side can resume from it (`train.py:61-72`).  `.pt` / `.pth` paths get the bare state dict
        (`train.py:105-116`)."""
        from . import checkpoint as ck
        torch.cuda.current_stream().synchronize()
        if str(path).endswith(".ckpt"):
            torch.save(ck.lightning_checkpoint(self.model, self, epoch), path)
        else:
            torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, path)

    def resume(self, path: str, strict: bool = False) -> int:
        """Load weights (and, from a `.ckpt`, optimizer moments and the step counter).  Returns the global
        step training continues from."""
        from . import checkpoint as ck
        blob = ck.read_checkpoint(path)
        self.model.load_state_dict(blob["state_dict"], strict=strict)
        step = 0
        if blob["optimizer"] is not None:
            order = ck.reference_parameter_order(self.model.cfg, self.model.segmem_num_layers)
            step = ck.adamw_state_to_flat(blob["optimizer"], self.flat, order)
            step = max(step, blob["global_step"])
        self.host_step = step
        self.step_dev.fill_(step)
        if blob["extra"]:
            self.engine.seed = int(blob["extra"]["dropout_seed"])
            self.engine._stream_ctr = int(blob["extra"]["dropout_stream_ctr"])
        if self.world > 1:
            dist.broadcast(self.flat.P, src=0)
            dist.broadcast(self.flat.M, src=0)
            dist.broadcast(self.flat.V, src=0)
        return step

    @torch.no_grad()
    def eval_loss(self, inputs, labels, targets_prev=None, audio: bool = False):
        self.model.eval()
        mel = self.mel_from_audio(inputs) if audio else inputs
        if self.engine.dt != torch.bfloat16:
            logits, _ = self.engine.forward(mel, labels, targets_prev, training=False, need_grad=False)
            return lib.cross_entropy(logits.view(-1, logits.shape[-1]), labels.reshape(-1), want_grad=False,
                                     weighted=self.weighted)[0]
        dec, _ = self.engine.forward(mel, labels, targets_prev, training=False, need_grad=False, want_logits=False)
        loss, _ = lib.lmhead_cross_entropy(dec, self.engine.W("lm_head"), labels.reshape(-1), want_grad=False,
                                           weighted=self.weighted)
        return loss
