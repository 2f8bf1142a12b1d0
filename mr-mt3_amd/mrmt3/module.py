"""nn.Module face of the MI355X engine: same constructor arguments, call signatures, `.config`
attributes and state-dict keys as the reference model classes, with the arithmetic
delegated to `mrmt3.engine.Engine`.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from .engine import Engine
from .params import FlatParams


def config_to_dict(config) -> dict:
    """Accepts a plain dict, an HF T5Config, an OmegaConf node or any attribute bag."""
    if isinstance(config, dict):
        return dict(config)
    if hasattr(config, "to_dict"):
        return dict(config.to_dict())
    try:
        from omegaconf import OmegaConf  # optional
        if OmegaConf.is_config(config):
            return dict(OmegaConf.to_container(config, resolve=True))
    except ImportError:
        pass
    return {k: getattr(config, k) for k in dir(config) if not k.startswith("_") and not callable(getattr(config, k))}


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state-dict keys."""


class _ModelFn(torch.autograd.Function):
    """Bridges the hand-written backward into torch.autograd so `loss.backward()` works when a
    caller (e.g. the Lightning `training_step` of the reference's tasks) computes the loss with
    torch ops on the returned logits.  Gradients are accumulated into the flat buffer and exposed
    through each parameter's `.grad` view; the function returns no tensor gradients."""

    @staticmethod
    def forward(ctx, anchor, model, mel, labels, targets_prev):
        logits, tape = model.engine.forward(mel, labels, targets_prev, training=model.training, need_grad=True)
        ctx.model, ctx.tape = model, tape
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        model.attach_grads()
        buckets = model._ddp_buckets()
        if buckets is None:
            model.engine.backward(ctx.tape, dlogits.contiguous(), on_layer_done=model._on_layer_done)
        else:
            # Several ranks and nobody else reduces these gradients (see MT3Module._ddp_buckets): do what torch DDP does
            # for the reference — average them over the ranks, overlapped with the rest of backward.
            buckets.reset()
            model.engine.backward(ctx.tape, dlogits.contiguous(), on_layer_done=buckets.on_layer_done)
            buckets.finish()
            model.flat.G.mul_(1.0 / buckets.world)
        ctx.tape = None
        return None, None, None, None, None


class MT3Module(nn.Module):
    VARIANT = "t5"

    def __init__(self, config, segmem_num_layers: int = 0, segmem_length: int = 64, compute_dtype=torch.bfloat16):
        super().__init__()
        cfg = config_to_dict(config)
        cfg.setdefault("num_decoder_layers", cfg.get("num_layers"))
        cfg.setdefault("layer_norm_epsilon", 1e-6)
        cfg.setdefault("decoder_start_token_id", 0)
        cfg.setdefault("pad_token_id", 0)
        cfg.setdefault("eos_token_id", 1)
        cfg.setdefault("dropout_rate", 0.1)
        if cfg.get("feed_forward_proj", "gated-gelu") != "gated-gelu":
            raise NotImplementedError("only the gated-gelu feed forward of MT3 is implemented")
        if cfg.get("tie_word_embeddings", False):
            raise NotImplementedError("tie_word_embeddings=True is not used by any MR-MT3 config")
        self.cfg = cfg
        self.config = SimpleNamespace(**cfg)          # `.config.eos_token_id` etc. (inference.py:208)
        self.model_dim = cfg["d_model"]
        self.segmem_num_layers = segmem_num_layers if self.VARIANT != "t5" else 0
        self.segmem_length = segmem_length
        self.flat = FlatParams(cfg, self.segmem_num_layers)
        self.engine = Engine(cfg, self.flat, self.VARIANT, segmem_length, self.segmem_num_layers, compute_dtype)
        self._anchor = torch.zeros(1, requires_grad=True)
        self._on_layer_done = None
        self._build_tree()
        self.reset_parameters()

    # ---- parameter tree with the reference's key schema ---------------------------------
    def _build_tree(self):
        self._views = {}
        for key in self.flat.shapes:
            p = nn.Parameter(self.flat.master(key))
            self._views[key] = p
            self._register(key, p)
        self.flat.version_sources = tuple(self._views.values())
        # aliases: the stacks' embed_tokens ARE proj / decoder_embed_tokens / segmem_proj (t5.py:64,70)
        self._register("encoder.embed_tokens.weight", self._views["proj.weight"])
        self._register("decoder.embed_tokens.weight", self._views["decoder_embed_tokens.weight"])
        stacks = ["encoder", "decoder"]
        if self.segmem_num_layers:
            self._register("segmem_encoder.embed_tokens.weight", self._views["segmem_proj.weight"])
            stacks.append("segmem_encoder")
        d = self.cfg["d_model"]
        for s in stacks:
            node = self._node(f"{s}.pos_emb")
            node.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d)))

    def _node(self, path):
        m = self
        for part in path.split("."):
            if part not in m._modules:
                m.add_module(part, _Node())
            m = m._modules[part]
        return m

    def _register(self, key, param):
        path, leaf = key.rsplit(".", 1)
        self._node(path).register_parameter(leaf, param)

    def reset_parameters(self, seed: int = 0):
        """T5-style init (the reference inherits HF `_init_weights`): see synthetic._std_for."""
        from .synthetic import _std_for
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for key, shp in self.flat.shapes.items():
                v = self.flat.master(key)
                if len(shp) == 1:
                    v.fill_(1.0)
                else:
                    v.copy_(torch.randn(shp, generator=g) * _std_for(key, self.cfg))

    def _apply(self, fn, recurse=True):
        """Move the flat buffers and re-point every Parameter at its slice (keeps q|k|v adjacency)."""
        self.flat.to(fn)
        for key, p in self._views.items():
            p.data = self.flat.master(key)
            if p.grad is not None:
                p.grad = None
        for m in self.modules():
            for name, buf in list(m._buffers.items()):
                if buf is not None:
                    m._buffers[name] = fn(buf)
        self._anchor = torch.zeros(1, requires_grad=True, device=self.flat.P.device)
        return self

    @property
    def device(self):
        return self.flat.P.device

    def _ddp_buckets(self):
        """The gradient exchange of the DROP-IN path (`loss.backward()` on the logits this module returned).

        The reference wraps its task in torch DDP (`pl.Trainer(strategy="ddp_find_unused_parameters_false")`,
        config/config.yaml:45, train.py:43-47).  The parameters of this module never enter the autograd graph (the
        hand-written backward fills the flat gradient buffer), so DDP's reducer has no hooks to fire: wrapped in DDP, or
        run under any multi-rank launcher without `mrmt3.trainer.Trainer`, every rank would silently keep its local
        gradients and the replicas would drift apart.  So whenever a process group with more than one rank is
        initialised and no `Trainer` owns the exchange (`_on_layer_done` unset), backward reduces the gradients itself
        with the same bucketed all-reduce the Trainer uses and averages them (DDP semantics).  MRMT3_DDP_AUTO=0 turns
        this off (then a multi-rank backward raises instead of diverging quietly)."""
        import torch.distributed as dist
        if self._on_layer_done is not None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        import os
        if os.environ.get("MRMT3_DDP_AUTO", "1") == "0":
            raise RuntimeError(
                "MR-MT3 MI355X module under a multi-rank process group without a gradient exchange: its gradients live in "
                "one flat buffer that torch DDP cannot see.  Use mrmt3.trainer.Trainer (bucketed RCCL all-reduce "
                "overlapped with backward) or leave MRMT3_DDP_AUTO=1 so that backward() averages them itself.")
        if getattr(self, "_auto_buckets", None) is None:
            from .ddp import GradBuckets
            b = GradBuckets(self.flat, self.cfg["num_layers"], self.cfg["num_decoder_layers"], self.segmem_num_layers > 0)
            b.before_fire = self.engine.join_wgrad
            b.producer_streams = lambda: [self.engine._side]
            self._auto_buckets = b
            # identical replicas to start from (DDP broadcasts rank 0's parameters at construction)
            dist.broadcast(self.flat.P, src=0)
        return self._auto_buckets

    def attach_grads(self):
        """Expose slices of the flat gradient buffer as `.grad` (zeroing it when grads were reset)."""
        G = self.flat.ensure_grads()
        first = next(iter(self._views.values()))
        if first.grad is None or first.grad.data_ptr() != self.flat.grad(next(iter(self._views))).data_ptr():
            G.zero_()
            for key, p in self._views.items():
                p.grad = self.flat.grad(key)

    def load_golden(self, seed: int = 365):
        from .synthetic import golden_weights
        with torch.no_grad():
            self.flat.load_numpy(golden_weights(self.cfg, self.segmem_num_layers, seed))
        return self

    # ---- reference call surface -------------------------------------------------------------------------
    def forward(self, inputs=None, labels=None, targets_prev=None, **ignored):
        """`forward(inputs, labels[, targets_prev]) -> lm_logits [B, Ld, V]` (models/t5.py:182-249)."""
        if inputs is None or labels is None:
            raise ValueError("forward needs inputs (mel) and labels")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._views.values()):
            self._ddp_buckets()        # several ranks and no Trainer: start from rank 0's weights (once), see there
            return _ModelFn.apply(self._anchor, self, inputs, labels, targets_prev)
        logits, _ = self.engine.forward(inputs, labels, targets_prev, training=self.training, need_grad=False)
        return logits

    def generate(self, inputs, max_length=1024, **kwargs):
        from .decode import generate
        return generate(self, inputs, max_length=max_length)
