"""Flat parameter store for the MR-MT3 models.

All weights of a model live in ONE contiguous fp32 buffer (master copy) laid out in state-dict
order, with matching flat buffers for gradients, AdamW moments and the bf16 shadow the MFMA kernels
read.  Consequences that matter on MI355X:
  * q|k|v, cross k|v and wi_0|wi_1 are adjacent, so the fused [1152,512] / [768,512] / [2048,512]
    GEMM operands are plain views — no concatenation, no copies;
  * AdamW is one kernel launch over the whole model (45.9 M / 48.5 M elements);
  * the data-parallel gradient exchange all-reduces contiguous slices of one buffer over RCCL.
`nn.Parameter`s handed to PyTorch (state_dict, optimizers, Lightning) are views into the master
buffer, so the reference's state-dict schema is preserved byte for byte.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import lib
from .synthetic import state_dict_shapes


class FlatParams:
    def __init__(self, cfg: dict, segmem_num_layers: int = 0, device="cpu"):
        self.cfg = cfg
        self.shapes = state_dict_shapes(cfg, segmem_num_layers)
        self.offsets: "OrderedDict[str, int]" = OrderedDict()
        off = 0
        # Flat order = state-dict order, except that the cross-attention k|v projections of ALL decoder layers sit
        # together in front of decoder block 0: they all multiply the same encoder output, so the forward projects it for
        # every layer in one [n_layers * 768, 512] GEMM and the backward returns its gradient in one GEMM with
        # K = n_layers * 768 ("decoder.ckv_all"); the per-layer [768, 512] views stay valid.
        n_dec = cfg["num_decoder_layers"]
        ckv = [f"decoder.block.{i}.layer.1.EncDecAttention.{n}.weight" for i in range(n_dec) for n in ("k", "v")]
        ckv_set = set(ckv)
        order = []
        for k in self.shapes:
            if k in ckv_set:
                continue
            if k == "decoder.block.0.layer.0.SelfAttention.q.weight":
                order.extend(ckv)
            order.append(k)
        assert len(order) == len(self.shapes)
        for k in order:
            self.offsets[k] = off
            n = 1
            for s in self.shapes[k]:
                n *= s
            off += n
        self.numel = off
        assert off % 4 == 0
        self.P = torch.zeros(off, dtype=torch.float32, device=device)   # master
        self.G = None          # gradients (lazily allocated)
        self.M = None          # AdamW exp_avg
        self.V = None          # AdamW exp_avg_sq
        self.S = None          # bf16 shadow of P
        self.ST = None         # pre-transposed bf16 weights for dgrad
        self._shadow_version = -1
        # tensors other than P whose in-place updates also change the master (the nn.Parameter views of
        # MT3Module: after `.to(device)` each owns a version counter of its own, so torch.optim / load_state_dict
        # writes do not bump P._version)
        self.version_sources = ()
        self._build_groups(segmem_num_layers)

    # ---- fused weight groups ------------------------------------------------------------------------
    def _build_groups(self, segmem_num_layers):
        cfg = self.cfg
        d, inner, dff = cfg["d_model"], cfg["d_kv"] * cfg["num_heads"], cfg["d_ff"]
        g: "OrderedDict[str, tuple]" = OrderedDict()   # name -> (offset, rows, cols)

        def add(name, first_key, rows, cols):
            g[name] = (self.offsets[first_key], rows, cols)

        def stack(prefix, n, dec):
            for i in range(n):
                b = f"{prefix}.block.{i}.layer"
                add(f"{prefix}.{i}.qkv", f"{b}.0.SelfAttention.q.weight", 3 * inner, d)
                add(f"{prefix}.{i}.o", f"{b}.0.SelfAttention.o.weight", d, inner)
                ff = 1
                if dec:
                    add(f"{prefix}.{i}.cq", f"{b}.1.EncDecAttention.q.weight", inner, d)
                    add(f"{prefix}.{i}.ckv", f"{b}.1.EncDecAttention.k.weight", 2 * inner, d)
                    add(f"{prefix}.{i}.co", f"{b}.1.EncDecAttention.o.weight", d, inner)
                    ff = 2
                add(f"{prefix}.{i}.wi", f"{b}.{ff}.DenseReluDense.wi_0.weight", 2 * dff, d)
                add(f"{prefix}.{i}.wo", f"{b}.{ff}.DenseReluDense.wo.weight", d, dff)

        add("proj", "proj.weight", d, d)
        stack("encoder", cfg["num_layers"], False)
        stack("decoder", cfg["num_decoder_layers"], True)
        if cfg["num_decoder_layers"] > 0:
            add("decoder.ckv_all", "decoder.block.0.layer.1.EncDecAttention.k.weight", cfg["num_decoder_layers"] * 2 * inner, d)
        add("lm_head", "lm_head.weight", cfg["vocab_size"], d)
        if segmem_num_layers:
            add("segmem_proj", "segmem_proj.weight", d, d)
            stack("segmem_encoder", segmem_num_layers, False)
        self.groups = g
        # pre-transposed (dgrad) copies: every group but proj (the mel input needs no gradient) and the per-layer
        # cross k|v views (their dgrad is the one K = n_layers * 768 product against "decoder.ckv_all")
        self.t_offsets = OrderedDict()
        off = 0
        for name, (_, r, c) in g.items():
            if name == "proj" or (name.startswith("decoder.") and name.endswith(".ckv")):
                continue
            self.t_offsets[name] = off
            off += r * c
        self.t_numel = off

    # ---- views ---------------------------------------------------------------------------------------
    def view(self, buf, key):
        shp = self.shapes[key]
        n = 1
        for s in shp:
            n *= s
        o = self.offsets[key]
        return buf[o:o + n].view(shp)

    def master(self, key):
        return self.view(self.P, key)

    def grad(self, key):
        return self.view(self.G, key)

    def W(self, name, dtype):
        """Fused weight [rows, cols] in the compute dtype."""
        o, r, c = self.groups[name]
        buf = self.S if dtype == torch.bfloat16 else self.P
        return buf[o:o + r * c].view(r, c)

    def WT(self, name):
        """Pre-transposed bf16 weight [cols, rows] (dgrad operand)."""
        _, r, c = self.groups[name]
        o = self.t_offsets[name]
        return self.ST[o:o + r * c].view(c, r)

    def GW(self, name):
        o, r, c = self.groups[name]
        return self.G[o:o + r * c].view(r, c)

    # ---- device / shadows --------------------------------------------------------------------------
    def to(self, fn):
        self.P = fn(self.P)
        for n in ("G", "M", "V"):
            b = getattr(self, n)
            if b is not None:
                setattr(self, n, fn(b))
        self.S = None
        self.ST = None
        self._shadow_version = -1
        if self.P.dtype != torch.float32:
            raise TypeError("MR-MT3 master weights stay fp32; pick the compute dtype on the model")

    def ensure_grads(self):
        if self.G is None or self.G.device != self.P.device:
            self.G = torch.zeros_like(self.P)
        return self.G

    def ensure_adam(self):
        if self.M is None or self.M.device != self.P.device:
            self.M = torch.zeros_like(self.P)
            self.V = torch.zeros_like(self.P)

    def master_version(self):
        """Changes whenever the master weights were written through P or through any registered view."""
        v = self.P._version
        for t in self.version_sources:
            v += t._version
        return v

    def refresh_shadows(self, force=False, need_transposed=True):
        """(Re)build the bf16 shadow and the transposed dgrad copies when the master changed."""
        if not self.P.is_cuda:
            raise RuntimeError("bf16 shadows live on the GPU (no CPU fallback)")
        ver = self.master_version()
        have_t = self.ST is not None
        if not force and self.S is not None and ver == self._shadow_version and (have_t or not need_transposed):
            return
        if self.S is None or self.S.device != self.P.device:
            self.S = torch.empty(self.numel, dtype=torch.bfloat16, device=self.P.device)
        if force or ver != self._shadow_version or self._shadow_version < 0:
            lib.cast(self.P, self.S)
        if need_transposed:
            self.refresh_transposed()
        self._shadow_version = ver

    def refresh_transposed(self):
        """Rebuild every pre-transposed dgrad weight from the bf16 shadow in ONE launch."""
        import numpy as np
        if self.ST is None or self.ST.device != self.P.device:
            self.ST = torch.empty(self.t_numel, dtype=torch.bfloat16, device=self.P.device)
            self._tr_tab = None
        if getattr(self, "_tr_tab", None) is None:
            rec, starts, tot = [], [], 0
            for name, (o, r, c) in self.groups.items():
                if name not in self.t_offsets:
                    continue
                rec.append((o, self.t_offsets[name], r, c))
                starts.append(tot)
                tot += ((r + 63) // 64) * ((c + 63) // 64)       # 64x64 tiles (csrc/rowops.hip TRB)
            tab = np.zeros(len(rec), dtype=[("src", "<i8"), ("dst", "<i8"), ("rows", "<i4"), ("cols", "<i4")])
            for i, t in enumerate(rec):
                tab[i] = t
            self._tr_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(self.P.device)
            self._tr_start = torch.tensor(starts, dtype=torch.int32, device=self.P.device)
            self._tr_n, self._tr_tiles = len(rec), tot
        lib.transpose_batched(self.S, self.ST, self._tr_tab, self._tr_start, self._tr_n, self._tr_tiles)

    def adamw_step(self, lr_dev, step_dev, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale=1.0):
        """torch.optim.AdamW semantics over the whole model in one launch; keeps shadows current."""
        self.ensure_adam()
        if self.S is None:
            self.refresh_shadows()
        lib.adamw_step(self.P, self.G, self.M, self.V, lr_dev, step_dev, betas[0], betas[1], eps, weight_decay,
                       grad_scale, shadow=self.S)
        self.refresh_transposed()
        self._shadow_version = self.master_version()

    def load_numpy(self, weights: dict):
        for k, v in weights.items():
            self.master(k).copy_(torch.from_numpy(v).to(self.P.device))
