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

Captured collectives (one graph with the all-reduces in-line; two graphs side by side with flag hand-offs) were built and
measured in round 5 and lost (+0.13-0.25 ms exposed / +3.2-3.8 ms: profiles/r05_collectives_ab.txt); round 6 took them out of
the product (profiles/tools/closed/trainer_captured_collectives_r5.py keeps the code for the record).

A capture that fails.  Whatever trips a capture (a kernel that refuses, an illegal call inside it), the step goes on with plain
launches — and nothing of the failed capture may stay behind, because of how torch 2.10 / ROCm 7.2 behave (round 6 root cause of
the round-5 abort, profiles/r06_capture_abort_root_cause.txt): (1) the cyclic garbage collector running INSIDE a capture frees
older trainers' graphs, page-locked tables and communicators — HIP calls that are illegal while the thread captures — and
invalidates the capture; (2) `CUDAGraph.capture_begin` on a stream that is still in capture mode raises AFTER the graph has noted
the default generator's state but BEFORE that state has noted the graph, and destroying such a graph object throws inside
`~CUDAGraph` -> std::terminate -> SIGABRT.  So: garbage is collected BEFORE a capture and the collector is off during it; a
capture never begins on a stream whose status is not "none" (mrmt3_stream_capture_status); after a failure every participating
stream is taken out of capture mode (mrmt3_stream_abandon_capture), the capture stream is replaced by a fresh one, and a graph
object whose capture_begin raised is never destroyed (`_retire`).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import lib
from .ddp import GradBuckets


_RETIRED = []


def _retire(graph) -> None:
    """A CUDAGraph whose capture_begin raised must never be destroyed: torch 2.10's ~CUDAGraph un-registers the graph from the
    generator state it noted in capture_begin, and when capture_begin raised between the two registrations that check throws
    inside the destructor and the process aborts.  One reference is leaked on purpose (a few hundred bytes)."""
    import ctypes
    _RETIRED.append(graph)
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(graph))


def _first_line(e) -> str:
    t = str(e)
    return t.splitlines()[0] if t else ""


class _CapturedStep:
    """One input signature's captured step: graph segments (each followed by the gradient buckets to send), the tail
    graph (AdamW) and the static tensors the graphs read and write."""

    def __init__(self):
        self.segments, self.tail = [], None
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
        self._collective_stream_checked = False
        self.graph_warmup = 2            # eager steps per input signature before capture (tables, workspaces)
        self._graphs = {}                # signature -> _CapturedStep
        self._eager_seen = {}
        self._cap_stream = None
        self._cap_owner = None

    def mel_from_audio(self, audio):
        """[B, n_samples] f32 device audio -> [B, frames, 512] mel in the compute dtype."""
        from contrib import spectrograms as sp
        return sp.logmel_segments(audio, out_bf16=(self.engine.dt == torch.bfloat16))

    # ---- one step's device work (identical in eager mode, under capture and — by replay — afterwards) ------------
    def _step_body(self, inputs, labels, targets_prev, audio, cut=None):
        """Enqueues one optimizer step.  `cut(bucket_indices)` is called where a gradient bucket is complete (only
        when collectives will run): under capture it closes the current graph segment."""
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
                cap = self._capture(sig, inputs, labels, targets_prev, audio)
            except Exception as e:     # noqa: BLE001 — whatever a capture trips over, the eager step is still correct
                # (nothing executed during the failed capture: the step below is the first to run; the launches the
                # aborted capture had deferred are dropped — _after_failed_capture and _step_body reset them)
                import warnings
                state = self._after_failed_capture(e)
                warnings.warn("hipGraph capture of the training step failed (%s: %s)%s; continuing with eager launches"
                              % (type(e).__name__, _first_line(e), state))
                self.use_graph = False
                return self._step_body(inputs, labels, targets_prev, audio)
        self.engine.prepare(True)          # weights written through torch since the last step? rebuild the shadows
        cap.inputs.copy_(inputs, non_blocking=True)
        cap.labels.copy_(labels, non_blocking=True)
        if cap.prev is not None:
            cap.prev.copy_(targets_prev, non_blocking=True)
        self.buckets.reset()
        for graph, fire in cap.segments:
            graph.replay()
            for idx in fire:
                self.buckets.fire(idx)
        self.buckets.wait()
        cap.tail.replay()
        return cap.loss.clone()            # the graph's own loss scalar is overwritten by the next replay

    # ---- a capture that failed: leave nothing behind -----------------------------------------------------------------
    def _capture_streams(self):
        """Every stream a capture of the step can have pulled into capture mode (a capture spreads to each stream that waits
        on an event of a capturing one: the capture stream itself, the engine's weight-gradient side stream, the collective
        stream and the candidates the stream pick made), plus the caller's."""
        out = {"current": torch.cuda.current_stream(), "capture": self._cap_stream, "side": self.engine._side,
               "collective": self.buckets._launch}
        for i, s in enumerate(getattr(self, "_stream_candidates", [])):
            out["candidate%d" % i] = s
        seen, uniq = set(), {}
        for k, v in out.items():
            if v is not None and v.cuda_stream not in seen:
                seen.add(v.cuda_stream)
                uniq[k] = v
        return uniq

    def _drop_capture_stream(self):
        """Forget the capture stream; if it is one of our own, destroy it (an invalidated stream is lost for capturing)."""
        owner, self._cap_owner, self._cap_stream = getattr(self, "_cap_owner", None), None, None
        if owner is not None:
            try:
                torch.cuda.synchronize()
            except RuntimeError:
                pass
            owner.close()

    def _after_failed_capture(self, exc) -> str:
        """Called with the exception of a capture that failed, BEFORE anything else is launched or synchronised.  Ends the
        capture on every stream that is still in capture mode (an exception between capture_begin and capture_end — or a
        capture_end that itself fails on an unjoined stream — leaves streams capturing; a device synchronise is illegal
        then), empties the thread's HIP error slot (the library's launch wrappers report whatever sits there as THEIR launch
        failure: one stale code would fail the retry and every eager launch after it), drops what the engine had deferred,
        then drains the device.  Returns a short state report for the warning; with MRMT3_CAPTURE_LOG=<file> the full report
        (traceback, per-stream capture status, live graph / stream / communicator counts) is appended there."""
        import gc
        import traceback
        left = []
        for name, st in self._capture_streams().items():
            was = lib.stream_abandon_capture(st)
            if was != "none":
                left.append("%s stream was left capturing (%s)" % (name, was))
        pending = lib.runtime_error_pop()
        if pending:
            left.append("pending HIP error %s" % pending)
        self._drop_capture_stream()                          # a later capture (another input shape) gets a fresh stream
        self.engine.reset_deferred()
        self.engine._stream_ctr = 0
        log = os.environ.get("MRMT3_CAPTURE_LOG")
        if log:
            objs = gc.get_objects()
            counts = dict(graphs=sum(isinstance(o, torch.cuda.CUDAGraph) for o in objs),
                          streams=sum(isinstance(o, torch.cuda.Stream) for o in objs),
                          comms=sum(isinstance(o, lib.Comm) for o in objs),
                          trainers=sum(isinstance(o, Trainer) for o in objs))
            with open(log, "a") as f:
                f.write("---- failed capture (world=%d, step=%d)\n%s%s\nlive objects: %s\n"
                        % (self.world, self.host_step,
                           "".join(traceback.format_exception(type(exc), exc, exc.__traceback__)),
                           "; ".join(left) or "no stream left capturing, no pending error", counts))
        try:
            torch.cuda.synchronize()
        except RuntimeError as e:        # a device that cannot be drained: say so in Python instead of going on blind
            raise RuntimeError("the device could not be synchronised after a failed graph capture (%s); state: %s"
                               % (_first_line(e), "; ".join(left) or "clean")) from exc
        return (" [" + "; ".join(left) + "]") if left else ""

    # ---- which stream the collectives run on --------------------------------------------------------------------------
    def _side_by_side(self, compute_stream, collective_stream, timeout_ms: int = 100) -> bool:
        """Do kernels of the two streams run side by side?  A spinning wait on the collective stream, then its signal on
        the compute stream: if the wait times out, both streams feed ONE hardware queue (HIP shares a few queues among
        the streams of a priority).  Eager, 0.1 ms when fine, `timeout_ms` when not."""
        dev = self.flat.G.device
        w = torch.zeros(3, dtype=torch.int32, device=dev)            # flag, seen, err
        torch.cuda.synchronize()
        lib.flag_wait(w[0:1], w[1:2], w[2:3], timeout_ms, stream=collective_stream)
        lib.flag_signal(w[0:1], stream=compute_stream)
        torch.cuda.synchronize()
        return int(w[2].item()) == 0

    def _pick_collective_stream(self, compute_stream, device) -> bool:
        """A collective stream on ANOTHER hardware queue than the compute stream's: on a shared queue an eager all-reduce
        simply queues between the backward kernels and overlaps nothing (profiles/r05_two_graph_probe.txt).  HIP deals the
        streams of one priority over a few queues, so: test the stream the buckets already use, then up to eight fresh ones
        (MRMT3_DDP_STREAM_PRIO: their priority, default normal — a resident kernel on a HIGH-priority queue slows the compute
        graph's launches more, profiles/r05_collectives_ab.txt) and keep the first that passes.  At most nine probes of
        100 ms: under a second in all.  False: none did."""
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

    def _capture(self, sig, inputs, labels, targets_prev, audio):
        """Record the step once (nothing executes during capture); `train_step` then replays it, this step included."""
        import gc
        eng = self.engine
        cur = torch.cuda.current_stream()
        if self._cap_stream is None:
            # a stream of our own, not one out of torch's pool: a stream that some failed capture left invalidated comes back
            # from that pool (round robin over 32), and ROCm never takes it out of capture mode again
            self._cap_owner = lib.OwnedStream(inputs.device)
            self._cap_stream = self._cap_owner.stream
        cs = self._cap_stream
        st = lib.stream_capture_status(cs)
        if st != "none":                   # (checked again before every segment's capture_begin: see begin())
            raise RuntimeError("the capture stream is still in capture mode (%s): not beginning another capture on it" % st)
        # Garbage first, while HIP calls are legal: an older trainer's graphs (hipGraphExecDestroy), page-locked plan tables
        # (hipHostFree) or communicator freed by the cyclic collector in the MIDDLE of the capture invalidate it ("operation
        # failed due to a previous error during capture" — seen only in long-lived processes with such garbage pending).
        gc.collect()
        torch.cuda.synchronize()           # nothing of the eager steps (collectives included) is in flight during capture
        eng.prepare(True)
        cap = _CapturedStep()
        cap.inputs, cap.labels = inputs.clone(), labels.clone()
        cap.prev = None if targets_prev is None else targets_prev.clone()
        cs.wait_stream(cur)
        pool = None
        state = {"g": None}

        def begin():
            # never on a stream that is not cleanly out of capture mode: torch's capture_begin would raise half-way through
            # its registrations and leave a graph object whose destructor aborts the process (module docstring)
            st = lib.stream_capture_status(cs)
            if st != "none":
                raise RuntimeError("the capture stream is still in capture mode (%s): not beginning another capture on it" % st)
            # thread-local capture mode: the process group's watchdog thread polls the events of earlier collectives
            # (hipEventQuery) whenever it likes; under the default global mode that call is illegal while ANY thread
            # captures and the watchdog takes the process down (seen with RCCL at world size 1, forced collectives)
            g = torch.cuda.CUDAGraph()
            try:
                if pool is None:
                    g.capture_begin(capture_error_mode="thread_local")
                else:
                    g.capture_begin(pool=pool, capture_error_mode="thread_local")
            except Exception:
                _retire(g)
                raise
            state["g"] = g

        def cut(fire):
            nonlocal pool
            g = state["g"]
            state["g"] = None
            g.capture_end()
            if pool is None:
                pool = g.pool()
            cap.segments.append((g, list(fire)))
            begin()

        overlap_was = eng.overlap_wgrad
        gc_was = gc.isenabled()
        gc.disable()
        try:
            if os.environ.get("MRMT3_GRAPH_LINEAR", "1") == "1":
                eng.overlap_wgrad = False          # one chain of nodes, no fork/join edges in the graph
            with torch.cuda.stream(cs):
                begin()
                try:
                    cap.loss = self._step_body(cap.inputs, cap.labels, cap.prev, audio, cut=cut)
                    g, state["g"] = state["g"], None
                    g.capture_end()
                    cap.tail = g
                except Exception:
                    g = state["g"]
                    if g is not None:              # a capture is open: end it here.  If it is already invalidated this raises
                        try:                       # too: _after_failed_capture then takes the stream out of capture mode, and
                            g.capture_end()        # the allocator is told by hand that this capture no longer routes
                        except Exception:          # allocations into its pool (capture_end raised before it got there)
                            try:
                                torch._C._cuda_endAllocateToPool(cap.inputs.device.index or 0, g.pool())
                            except Exception:
                                pass
                    raise
        finally:
            eng.overlap_wgrad = overlap_was
            if gc_was:
                gc.enable()
        cur.wait_stream(cs)
        self._graphs[sig] = cap
        return cap

    def close(self):
        """Release what the trainer holds on the device in an order that is safe: drain, drop the captured graphs (their
        executables are destroyed now, not by the garbage collector at some later HIP-illegal moment), then the library's
        communicator if the buckets made one.  The trainer is unusable for graph replay afterwards; eager steps still work."""
        import gc
        torch.cuda.synchronize()
        self._graphs.clear()
        self._eager_seen.clear()
        gc.collect()
        torch.cuda.synchronize()
        self._drop_capture_stream()
        self.buckets.close()

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
