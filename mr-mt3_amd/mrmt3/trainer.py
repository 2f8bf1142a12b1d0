"""Training step of the MI355X path: audio -> log-mel -> forward -> fused CE -> hand-written backward
(overlapped with the RCCL gradient exchange) -> one-launch AdamW.

This is the build's counterpart of what `train.py:43-103` gets from `pl.Trainer.fit` +
`MT3Net*.training_step` + `configure_optimizers` (tasks/mt3_net.py:27-68): same loss, same
optimizer semantics (torch.optim.AdamW defaults, cosine-warmup LambdaLR stepped per batch), one
process per GPU.  Nothing on the host waits for the device inside a step: the learning rate, the
step counter and the loss stay in device memory.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import lib
from .ddp import GradBuckets


class Trainer:
    def __init__(self, model, lr: float = 2e-4, lr_lambda=None, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, weighted_loss: bool = False, layers_per_bucket: int = 2):
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
        self.buckets = GradBuckets(self.flat, cfg["num_layers"], cfg["num_decoder_layers"],
                                   model.segmem_num_layers > 0, layers_per_bucket)
        self.buckets.before_fire = model.engine.flush_norm_dw
        self.buckets.producer_streams = lambda: [model.engine._side]
        self.flat.ensure_grads()
        self.flat.ensure_adam()
        if self.world > 1:   # C2: identical replicas
            dist.broadcast(self.flat.P, src=0)
        self.last_loss = None

    def mel_from_audio(self, audio):
        """[B, n_samples] f32 device audio -> [B, frames, 512] mel in the compute dtype."""
        from contrib import spectrograms as sp
        return sp.logmel_segments(audio, out_bf16=(self.engine.dt == torch.bfloat16))

    def train_step(self, inputs, labels, targets_prev=None, audio: bool = False):
        """One optimizer step.  `inputs` is mel [B,Le,512] or, with audio=True, raw audio [B,n].
        Returns the (device, un-synchronised) mean loss of this rank."""
        m, eng, flat = self.model, self.engine, self.flat
        m.train()
        if self.lr_lambda is not None:
            self.lr_dev.fill_(self.base_lr * self.lr_lambda(self.host_step))
        mel = self.mel_from_audio(inputs) if audio else inputs
        logits, tape = eng.forward(mel, labels, targets_prev, training=True, need_grad=True)
        B, Ld, V = logits.shape
        loss, dl = lib.cross_entropy(logits.view(B * Ld, V), labels.reshape(-1), want_grad=True,
                                     grad_dtype=torch.bfloat16, weighted=self.weighted)
        del logits
        flat.G.zero_()
        self.buckets.reset()
        eng.backward(tape, dl, on_layer_done=self.buckets.on_layer_done)
        self.buckets.finish()
        flat.adamw_step(self.lr_dev, self.step_dev, self.betas, self.eps, self.wd, grad_scale=1.0 / self.world)
        self.host_step += 1
        if self.world > 1:   # C4: logged loss, reduced without blocking the host
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, async_op=True).wait()   # stream-level wait only
            loss /= self.world
        self.last_loss = loss
        return loss

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
        logits, _ = self.engine.forward(mel, labels, targets_prev, training=False, need_grad=False)
        B, Ld, V = logits.shape
        loss, _ = lib.cross_entropy(logits.view(B * Ld, V), labels.reshape(-1), want_grad=False, weighted=self.weighted)
        return loss
