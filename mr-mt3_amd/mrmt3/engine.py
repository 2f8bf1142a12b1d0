"""Forward / backward of the MR-MT3 encoder–decoder as an explicit tape over the C-ABI kernels.

This is the host-side wiring that the reference gets from HF `T5Block` + autograd
(`models/t5.py:99-180,478-702`, `models/t5_segmem*.py`): every sublayer is a short, fixed sequence
of kernel launches, and the backward is written out by hand in reverse order so that
  * activations are kept in the dtype the next kernel consumes (bf16 GEMM operands, fp32 residual
    stream and statistics),
  * weight gradients accumulate straight into the flat gradient buffer (`FlatParams.G`),
  * a per-layer callback can start the RCCL all-reduce of finished gradient slices while the rest of
    the backward still runs (`on_layer_done`).

Residual-stream convention: a sublayer returns its un-dropped output `y`; the NEXT fused kernel
(`add_rmsnorm`) applies dropout to `y`, adds it to the residual `x` and normalises — one pass over
the row for three reference ops (`hidden + dropout(sublayer)`, then `T5LayerNorm`).
"""
from __future__ import annotations

import os

import torch

from . import lib
from .params import FlatParams
from .synthetic import sinusoid_table


class Tape:
    """Saved tensors of one forward pass (dropped as soon as backward has consumed them)."""

    def __init__(self):
        self.ops = []

    def push(self, **kw):
        self.ops.append(kw)

    def pop(self):
        return self.ops.pop()


class Engine:
    def __init__(self, cfg: dict, flat: FlatParams, variant: str = "t5", segmem_length: int = 64,
                 segmem_num_layers: int = 0, compute_dtype=torch.bfloat16, max_pos: int = 5000):
        self.cfg, self.flat, self.variant = cfg, flat, variant
        self.segmem_length, self.segmem_num_layers = segmem_length, segmem_num_layers
        self.dt = compute_dtype
        self.d, self.H, self.dff, self.V = cfg["d_model"], cfg["num_heads"], cfg["d_ff"], cfg["vocab_size"]
        self.inner = cfg["d_kv"] * self.H
        assert cfg["d_kv"] == 64, "kernels are specialised for d_kv = 64"
        self.eps = cfg["layer_norm_epsilon"]
        self.max_pos = max_pos
        self._pos = None
        self._stream_ctr = 0
        self.seed = 365
        # optional DEVICE step counter (int32[1]) salting every dropout mask in-kernel.  The trainer sets it (and
        # restarts the site counter every step) so that a hipGraph replay, whose by-value arguments are frozen,
        # still draws fresh masks each step — and draws the SAME masks as the eager path would.
        self.step_dev = None
        # dtype of sublayer outputs / dgrad outputs entering the fp32 residual add: the compute dtype
        # (bf16 halves that stream's HBM traffic; the residual itself and all statistics stay fp32)
        self.y_dtype = compute_dtype
        # weight-gradient GEMMs are off the critical dgrad chain (their outputs are only needed by the optimizer / the
        # gradient exchange): the shapes the grouped kernel takes are deferred to join_wgrad() (tn_group below); the
        # others run one by one on a second HIP stream and overlap the attention / dgrad kernels of the layers below
        self.overlap_wgrad = os.environ.get("MRMT3_WGRAD_STREAM", "1") != "0"
        # residual-stream GRADIENT of the bf16 path: bf16 between the norm-backward sites of a stack (f32 at both
        # ends) halves the largest streams of the backward row kernels; the forward residual stays f32
        self.res_grad_dtype = (torch.bfloat16 if compute_dtype == torch.bfloat16 and
                               os.environ.get("MRMT3_RES_GRAD", "bf16") == "bf16" else torch.float32)
        # lm_head (and the final decoder norm feeding it) in "bf16" (MFMA bf16 operands) or "f32" (exact-f32 MFMA on
        # the master weights; inference / parity only)
        self.head_dtype = "bf16"
        self._wt32 = {}
        self._side = None
        self._side_dirty = False
        self._held = []
        # bf16(O - bf16(O)) of an attention site is kept for the backward's delta where the value rows share a large common
        # component — cross-attention over near-identical encoder frames.  "all": every site; "cross": the
        # self-attention sites skip the extra 50 MB write + read per site
        self.lo_sites = os.environ.get("MRMT3_ATTN_LO", "all")
        self.norm_dw = lib.NormDwBatch() if os.environ.get("MRMT3_NORM_DW_BATCH", "1") != "0" else None
        # split-K slabs of the weight-gradient GEMMs: kept per site and summed in one launch (lib.TnBatch)
        self.tn_batch = lib.TnBatch() if os.environ.get("MRMT3_TN_BATCH", "1") != "0" else None
        if self.tn_batch is not None:
            self.tn_batch.before_early_flush = self._join_side
        # ... and, for every shape the grouped kernel takes, the GEMMs themselves are deferred to the next join_wgrad():
        # one launch for the weight gradients of a whole gradient bucket (lib.TnGroup)
        self.tn_group = lib.TnGroup() if os.environ.get("MRMT3_TN_GROUP", "1") != "0" else None
        # projection + row kernel in one launch (csrc/gemm_rows.hip; bf16 engine only).  MRMT3_FUSE_ROWS bits: 1 the o / co /
        # wo projection with the residual add + norm behind it, 2 the 512-column data gradients with the norm backward,
        # 4 the wo data gradient with the gated-GELU backward.  Same results as the two-kernel form (tests/test_gemm_rows_gpu.py).
        # Default 6: same-box step A/B at 64 segments (profiles/r04_fuse_rows_step_ab.txt) 24.68 ms unfused, 24.25-24.29 with
        # bit 4, 24.18 with bits 2 + 4 (K <= 1152), 24.45-24.47 when bit 1 is added — the forward fusion costs what it saves
        # (its K loop is bound by the CU's L2 -> LDS rate and does not overlap its row phase)
        self.fuse_rows = int(os.environ.get("MRMT3_FUSE_ROWS", "6")) if compute_dtype == torch.bfloat16 else 0
        # the norm-backward fusion up to this K (d_cq 384, d_qkv 1152; d_wi, K = 2048, stays on the ping-pong product +
        # the stand-alone row kernel: 224 against 205 us cold, profiles/r04_gemm_rows_ab.txt)
        self.fuse_normbwd_max_k = int(os.environ.get("MRMT3_FUSE_NORMBWD_MAXK", "1152"))

    # ---- helpers ---------------------------------------------------------------------------------------
    def pos(self, device):
        if self._pos is None or self._pos.device != device:
            self._pos = sinusoid_table(self.max_pos, self.d).to(device).contiguous()
        return self._pos

    def _sid(self):
        self._stream_ctr += 1
        return self._stream_ctr

    def W(self, name):
        return self.flat.W(name, self.dt)

    def ln(self, key):
        return self.flat.master(key)

    def WT(self, name):
        """Transposed weight [in, out] for the data-gradient products: the pre-transposed bf16 copy, or — fp32 path — an
        f32 transpose of the master made on demand (kept for one forward/backward pass: prepare() drops them, the AdamW
        kernel rewrites the master without touching torch's version counters)."""
        if self.dt == torch.bfloat16:
            return self.flat.WT(name)
        t = self._wt32.get(name)
        if t is None:
            w = self.flat.W(name, torch.float32)
            t = torch.empty(w.shape[1], w.shape[0], device=w.device, dtype=torch.float32)
            lib.transpose(w, t)
            self._wt32[name] = t
        return t

    def side_stream(self):
        if self._side is None or self._side.device != torch.cuda.current_stream().device:
            self._side = torch.cuda.Stream(priority=int(os.environ.get("MRMT3_WGRAD_PRIO", "-1")))
            self._events = [torch.cuda.Event() for _ in range(64)]
            self._ev_i = 0
        return self._side

    def wgrad(self, a, b, out):
        """out += a^T @ b, complete after the next join_wgrad() (inputs were produced on the current stream).
        Grouped path: only recorded here (operands kept alive), launched with every other gradient that is due.
        One-by-one path (shapes the grouped kernel does not take): on the side stream, kept cheap on the host — events
        come from a small ring, the kernel is launched on the side stream directly, and the operands are kept alive by
        reference until the next join instead of `record_stream` bookkeeping."""
        if a.dtype == torch.float32:                      # fp32 training / parity path: exact-f32 product, in place, now
            return lib.gemm_tn_f32(a, b, out, accumulate=True)
        if self.tn_group is not None and self.tn_group.ok(a, b, out):
            return self.tn_group.add(a, b, out, accumulate=True)
        if not self.overlap_wgrad:
            return lib.gemm_tn(a, b, out, accumulate=True, defer=self.tn_batch)
        side = self.side_stream()
        ev = self._events[self._ev_i]
        self._ev_i = (self._ev_i + 1) % len(self._events)
        ev.record()
        side.wait_event(ev)
        lib.gemm_tn(a, b, out, accumulate=True, stream=side, defer=self.tn_batch)
        self._held.append((a, b))
        self._side_dirty = True

    def _norm_bwd(self, *a, **kw):
        """lib.add_rmsnorm_bwd with the norm-weight gradient deferred to one batched reduction (flush_norm_dw).
        fp32 path: the kernel's second output (the masked gradient of the sublayer output below) is bf16-only, so it is
        formed from dx1 here — dx1 itself without dropout, one mask-and-copy launch with it."""
        if self.dt == torch.bfloat16:
            return lib.add_rmsnorm_bwd(*a, defer=self.norm_dw, **kw)
        want_dy = kw.pop("want_dy", True)
        dx1, _ = lib.add_rmsnorm_bwd(*a, defer=self.norm_dw, want_dy=False, **kw)
        if not want_dy:
            return dx1, None
        p = kw.get("p", 0.0)
        if p <= 0.0:
            return dx1, dx1
        return dx1, lib.dropmask_cast(dx1, p=p, seed=kw.get("seed", 0), stream_id=kw.get("stream_y", 0),
                                      step=kw.get("step"), out_dtype=torch.float32)

    def _proj_addnorm(self, x, pend, w_norm, xn_dtype=None, **kw):
        """add_rmsnorm_fwd(x, y, ...) where y is the pending projection `pend` = (a, W) of the sublayer above (None: the
        stack's first norm, nothing to add): one fused launch when the shape allows, else the product then the row kernel."""
        xn_dtype = xn_dtype or self.dt
        if pend is None:
            return lib.add_rmsnorm_fwd(x, None, w_norm, self.eps, xn_dtype, **kw)
        a, w = pend
        if ((self.fuse_rows & 1) and self.y_dtype == torch.bfloat16 and xn_dtype == torch.bfloat16 and x.shape[1] == 512 and
                lib.gemm_rows_ok(a, w)):
            return lib.gemm_nt_addnorm(a, w, x, w_norm, self.eps, **kw)
        y = lib.gemm_nt(a, w, out_dtype=self.y_dtype)
        return lib.add_rmsnorm_fwd(x, y, w_norm, self.eps, xn_dtype, **kw)

    def _proj_norm_bwd(self, a, wt, dres, x1, rstd, w_norm, dw, **kw):
        """_norm_bwd(gemm_nt(a, wt), dres, ...): the data gradient of a sublayer's input projection and the backward of the
        norm in front of it, one fused launch when the shape allows."""
        # (only with the batched norm-weight reduce: without it the fused call would have to build its one-site reduce table
        # by a host-to-device copy, which the capture of the step does not allow — ADVICE r4)
        if ((self.fuse_rows & 2) and self.norm_dw is not None and self.y_dtype == torch.bfloat16 and dres is not None and
                x1.shape[1] == 512 and a.shape[1] <= self.fuse_normbwd_max_k and lib.gemm_rows_ok(a, wt)):
            return lib.gemm_nt_normbwd(a, wt, dres, x1, rstd, w_norm, dw, defer=self.norm_dw, **kw)
        dxn = lib.gemm_nt(a, wt, out_dtype=self.y_dtype)
        return self._norm_bwd(dxn, dres, x1, rstd, w_norm, dw, **kw)

    def _proj_geglu_bwd(self, dy, wt, h, **kw):
        """geglu_bwd(h, gemm_nt(dy, wt)): the wo data gradient and the gated-GELU backward."""
        if (self.fuse_rows & 4) and h.dtype == torch.bfloat16 and lib.gemm_rows_ok(dy, wt, N=wt.shape[0]):
            return lib.gemm_nt_geglubwd(dy, wt, h, **kw)
        dg = lib.gemm_nt(dy, wt)
        return lib.geglu_bwd(h, dg, **kw)

    def flush_norm_dw(self):
        if self.norm_dw is not None:
            self.norm_dw.flush()

    def _join_side(self):
        """The current stream waits for the side stream's weight-gradient launches issued so far."""
        if self.overlap_wgrad and self._side is not None and self._side_dirty:
            # (only when the side stream has work since the last join: under graph capture a wait on a stream that
            # is not part of the capture would tie the graph to uncaptured work)
            self._side_dirty = False
            torch.cuda.current_stream().wait_stream(self._side)
            # operands may be recycled now: whatever the current stream does next runs after the side work
            self._held.clear()

    def join_wgrad(self):
        """Make the current stream wait for every weight gradient issued so far (and sum the queued norm-weight
        gradients: both are what a gradient bucket needs before it is sent)."""
        self.flush_norm_dw()
        self._join_side()
        if self.tn_batch is not None:
            self.tn_batch.flush()        # behind the GEMMs that produced the slabs (same stream, or joined above)
        if self.tn_group is not None:
            self.tn_group.flush()        # the deferred weight gradients: their operands were produced on this stream

    def reset_deferred(self):
        """Forget every deferred launch: the weight gradients recorded for the grouped launch, the queued split-K slab
        and norm-weight reductions, the operands held for the side stream.  After an aborted hipGraph capture (or an
        exception in the middle of a backward pass) these still name tensors of work that never ran — a later flush
        would accumulate uninitialised memory into the gradients."""
        if self.tn_group is not None:
            self.tn_group._sites.clear()
        if self.tn_batch is not None:
            self.tn_batch._queue.clear()
        if self.norm_dw is not None:
            self.norm_dw._queue.clear()
        self._held.clear()
        self._side_dirty = False

    def prepare(self, training: bool):
        if self.dt == torch.bfloat16:
            self.flat.refresh_shadows(need_transposed=training)
        else:
            self._wt32 = {}

    # ---- one T5Stack (models/t5.py:507-702) ----------------------------------------------------------
    def stack_fwd(self, prefix, x, B, L, n_layers, is_decoder, enc=None, Le=0, p=0.0, tape=None, out_dtype=None):
        """x: [B*L, d] fp32 (embeddings + sinusoid, dropout already applied).
        enc: [B*Le, d] compute-dtype encoder states for cross-attention.  Returns the final-normed
        (and dropped) states [B*L, d] in the compute dtype."""
        f, dt, H, inner, eps = self.flat, self.dt, self.H, self.inner, self.eps
        keep = tape is not None
        pend = None          # the projection of the sublayer above, not yet added to the residual stream: (operand, weight)
        sy = 0
        # cross-attention K | V of every layer in one projection of the encoder output: [B*Le, n_layers * 768]
        kv_all = lib.gemm_nt(enc, self.W(f"{prefix}.ckv_all")) if is_decoder and n_layers > 0 else None
        for i in range(n_layers):
            b = f"{prefix}.block.{i}.layer"
            # -- self attention
            s_in = sy
            x, xn, rstd = self._proj_addnorm(x, pend, self.ln(f"{b}.0.layer_norm.weight"), write_x1=True,
                                             p=p, seed=self.seed, step=self.step_dev, stream_y=s_in, x1=None if keep else x)
            qkv = lib.gemm_nt(xn, self.W(f"{prefix}.{i}.qkv"))
            s_att = self._sid()
            o, lse, o_lo = lib.attn_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, H, L, L,
                                        is_decoder, p=p, seed=self.seed, step=self.step_dev, stream_id=s_att,
                                        want_lse=keep, want_lo=keep and self.lo_sites == "all")
            pend = (o, self.W(f"{prefix}.{i}.o"))
            sy = self._sid()
            if keep:
                tape.push(kind="self", i=i, x1=x, xn=xn, rstd=rstd, qkv=qkv, o=o, o_lo=o_lo, lse=lse, s_in=s_in,
                          s_att=s_att)
            ff = 1
            if is_decoder and enc is not None:
                s_in = sy
                x, xn, rstd = self._proj_addnorm(x, pend, self.ln(f"{b}.1.layer_norm.weight"),
                                                 p=p, seed=self.seed, step=self.step_dev, stream_y=s_in, x1=None if keep else x)
                q = lib.gemm_nt(xn, self.W(f"{prefix}.{i}.cq"))
                kv = kv_all[:, i * 2 * inner:(i + 1) * 2 * inner]        # this layer's K | V columns (row stride n_layers * 768)
                s_att = self._sid()
                o, lse, o_lo = lib.attn_fwd(q, kv[:, :inner], kv[:, inner:], B, H, L, Le, False, p=p, seed=self.seed,
                                            step=self.step_dev, stream_id=s_att, want_lse=keep, want_lo=keep)
                pend = (o, self.W(f"{prefix}.{i}.co"))
                sy = self._sid()
                if keep:
                    tape.push(kind="cross", i=i, x1=x, xn=xn, rstd=rstd, q=q, kv=kv, o=o, o_lo=o_lo, lse=lse,
                              s_in=s_in, s_att=s_att)
                ff = 2
            # -- gated-GELU feed forward
            s_in = sy
            x, xn, rstd = self._proj_addnorm(x, pend, self.ln(f"{b}.{ff}.layer_norm.weight"),
                                             p=p, seed=self.seed, step=self.step_dev, stream_y=s_in, x1=None if keep else x)
            s_g = self._sid()
            if xn.dtype == torch.bfloat16:         # K2 + K7 in one launch (same bits as the two kernels)
                h, g = lib.gemm_nt_geglu(xn, self.W(f"{prefix}.{i}.wi"), p=p, seed=self.seed, step=self.step_dev, stream_id=s_g)
            else:
                h = lib.gemm_nt(xn, self.W(f"{prefix}.{i}.wi"))
                g = lib.geglu_fwd(h, p=p, seed=self.seed, step=self.step_dev, stream_id=s_g)
            pend = (g, self.W(f"{prefix}.{i}.wo"))
            sy = self._sid()
            if keep:
                tape.push(kind="ff", i=i, ff=ff, x1=x, xn=xn, rstd=rstd, h=h, g=g, s_in=s_in, s_g=s_g)
        s_out = self._sid()
        x, out, rstd = self._proj_addnorm(x, pend, self.ln(f"{prefix}.final_layer_norm.weight"), xn_dtype=out_dtype or dt, p=p,
                                          seed=self.seed, step=self.step_dev, stream_y=sy, stream_out=s_out, out_drop=True,
                                          x1=None if keep else x)
        if keep:
            tape.push(kind="final", x1=x, rstd=rstd, s_in=sy, s_out=s_out, prefix=prefix, n_layers=n_layers,
                      is_decoder=is_decoder, B=B, L=L, Le=Le, enc=enc, p=p)
        return out

    def stack_bwd(self, tape, d_out, d_enc=None, on_layer_done=None):
        """d_out: fp32 [B*L, d] gradient w.r.t. the stack output.  Accumulates weight grads into
        flat.G, adds cross-attention gradients into d_enc (fp32 [B*Le, d]) and returns the gradient
        w.r.t. the stack input x."""
        f = self.flat
        fin = tape.pop()
        assert fin["kind"] == "final"
        prefix, n_layers, is_dec = fin["prefix"], fin["n_layers"], fin["is_decoder"]
        B, L, Le, enc, p = fin["B"], fin["L"], fin["Le"], fin["enc"], fin["p"]
        H, inner, seed = self.H, self.inner, self.seed
        has_y = n_layers > 0
        rg = self.res_grad_dtype if has_y else torch.float32
        dx, dy = self._norm_bwd(d_out, None, fin["x1"], fin["rstd"], self.ln(f"{prefix}.final_layer_norm.weight"),
                                     f.grad(f"{prefix}.final_layer_norm.weight"), want_dy=has_y, p=p, seed=seed, step=self.step_dev,
                                     stream_y=fin["s_in"], stream_out=fin["s_out"], out_drop=True, dx1_dtype=rg)
        # gradient of the cross-attention K | V of every layer, side by side like the forward's kv_all: the gradient of
        # the encoder output is then ONE product with K = n_layers * 768 after the loop instead of n_layers f32 accumulations
        dkv_all = (torch.empty(B * Le, n_layers * 2 * inner, device=d_out.device, dtype=self.dt)
                   if is_dec and n_layers > 0 else None)
        for i in reversed(range(n_layers)):
            b = f"{prefix}.block.{i}.layer"
            t = tape.pop()
            assert t["kind"] == "ff" and t["i"] == i
            ff = t["ff"]
            self.wgrad(dy, t["g"], f.GW(f"{prefix}.{i}.wo"))
            dh = self._proj_geglu_bwd(dy, self.WT(f"{prefix}.{i}.wo"), t["h"], p=p, seed=seed, step=self.step_dev,
                                      stream_id=t["s_g"])
            self.wgrad(dh, t["xn"], f.GW(f"{prefix}.{i}.wi"))
            dx, dy = self._proj_norm_bwd(dh, self.WT(f"{prefix}.{i}.wi"), dx, t["x1"], t["rstd"],
                                         self.ln(f"{b}.{ff}.layer_norm.weight"), f.grad(f"{b}.{ff}.layer_norm.weight"),
                                         p=p, seed=seed, step=self.step_dev, stream_y=t["s_in"], dx1=dx)
            if ff == 2:
                t = tape.pop()
                assert t["kind"] == "cross"
                self.wgrad(dy, t["o"], f.GW(f"{prefix}.{i}.co"))
                do = lib.gemm_nt(dy, self.WT(f"{prefix}.{i}.co"))
                dq = torch.empty_like(t["q"])
                dkv = dkv_all[:, i * 2 * inner:(i + 1) * 2 * inner]
                kv = t["kv"]
                lib.attn_bwd(t["q"], kv[:, :inner], kv[:, inner:], t["o"], do, t["lse"], dq, dkv[:, :inner],
                             dkv[:, inner:], B, H, L, Le, False, p=p, seed=seed, step=self.step_dev,
                             stream_id=t["s_att"], o_lo=t["o_lo"])
                self.wgrad(dq, t["xn"], f.GW(f"{prefix}.{i}.cq"))
                self.wgrad(dkv, enc, f.GW(f"{prefix}.{i}.ckv"))
                dx, dy = self._proj_norm_bwd(dq, self.WT(f"{prefix}.{i}.cq"), dx, t["x1"], t["rstd"],
                                             self.ln(f"{b}.1.layer_norm.weight"), f.grad(f"{b}.1.layer_norm.weight"),
                                             p=p, seed=seed, step=self.step_dev, stream_y=t["s_in"], dx1=dx)
            t = tape.pop()
            assert t["kind"] == "self" and t["i"] == i
            self.wgrad(dy, t["o"], f.GW(f"{prefix}.{i}.o"))
            do = lib.gemm_nt(dy, self.WT(f"{prefix}.{i}.o"))
            qkv = t["qkv"]
            dqkv = torch.empty_like(qkv)
            lib.attn_bwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], t["o"], do, t["lse"],
                         dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:], B, H, L, L, is_dec, p=p,
                         seed=seed, step=self.step_dev, stream_id=t["s_att"], o_lo=t["o_lo"])
            self.wgrad(dqkv, t["xn"], f.GW(f"{prefix}.{i}.qkv"))
            last = i == 0                                     # the stack's input gradient leaves in f32
            dx, dy = self._proj_norm_bwd(dqkv, self.WT(f"{prefix}.{i}.qkv"), dx, t["x1"], t["rstd"],
                                         self.ln(f"{b}.0.layer_norm.weight"), f.grad(f"{b}.0.layer_norm.weight"),
                                         want_dy=(i > 0), p=p, seed=seed, step=self.step_dev, stream_y=t["s_in"],
                                         dx1=None if (last and dx.dtype != torch.float32) else dx)
            if on_layer_done is not None:
                on_layer_done(prefix, i)
        if dkv_all is not None:
            lib.gemm_nt(dkv_all, self.WT(f"{prefix}.ckv_all"), out=d_enc)          # d_enc is written here and nowhere else
        return dx

    # ---- model pieces ------------------------------------------------------------------------------------
    def _act(self, t):
        """torch tensor -> contiguous compute-dtype 2-D operand."""
        t = t.contiguous()
        if t.dtype != self.dt:
            out = torch.empty(t.shape, device=t.device, dtype=self.dt)
            lib.cast(t.view(-1), out.view(-1))
            t = out
        return t

    def encode(self, mel, p=0.0, tape=None):
        """proj -> + sinusoid -> dropout -> encoder stack (models/t5.py:122-136).  mel [B,Le,d]."""
        B, Le, d = mel.shape
        mel2 = self._act(mel.reshape(B * Le, d))
        src = lib.gemm_nt(mel2, self.W("proj"), out_dtype=self.y_dtype)
        s_emb = self._sid()
        x = lib.addpos_fwd(src, self.pos(mel.device), Le, p=p, seed=self.seed, step=self.step_dev, stream_id=s_emb)
        if tape is not None:
            tape.push(kind="enc_in", mel=mel2, s_emb=s_emb, p=p)
        return self.stack_fwd("encoder", x, B, Le, self.cfg["num_layers"], False, p=p, tape=tape)

    def encode_bwd(self, tape, d_enc_out, on_layer_done=None):
        dx = self.stack_bwd(tape, d_enc_out, on_layer_done=on_layer_done)
        t = tape.pop()
        assert t["kind"] == "enc_in"
        dsrc = lib.dropmask_cast(dx, p=t["p"], seed=self.seed, step=self.step_dev, stream_id=t["s_emb"], out_dtype=self.dt)
        self.wgrad(dsrc, t["mel"], self.flat.GW("proj"))

    def segmem(self, ids, B, L, tape=None):
        """embed -> segmem_proj -> 1-layer bidirectional encoder, dropout 0
        (models/t5_segmem.py:56-66, models/t5_segmem_v2_with_prev.py:121-123; the positional call
        `self.segmem_encoder(segmem_embeds)` routes the embeddings through `embed_tokens` =
        segmem_proj).  Returns the first `segmem_length` positions [B, Ls, d] in the compute dtype.

        Only those Ls outputs are ever used (`[:, :self.segmem_length]`), so for the reference's
        one-layer memory encoder the computation is restricted EXACTLY: keys/values need all L
        positions, but queries, the O projection, the feed-forward and the final norm only the first
        Ls rows (6.4 -> ~1 GFLOP per segment, SURVEY §7 "segmem-encoder shortcut")."""
        f, dt, d, H, inner, eps = self.flat, self.dt, self.d, self.H, self.inner, self.eps
        Ls = min(self.segmem_length, L)
        emb = lib.embed_fwd(ids.reshape(-1), f.master("decoder_embed_tokens.weight"), None, L, shift=False,
                            pad_id=self.cfg["pad_token_id"])
        emb_a = self._act(emb)
        src = lib.gemm_nt(emb_a, self.W("segmem_proj"), out_dtype=self.y_dtype)
        x = lib.addpos_fwd(src, self.pos(ids.device), L)
        if self.segmem_num_layers != 1:
            if tape is not None:
                tape.push(kind="seg_in", ids=ids.reshape(-1), emb=emb_a, L=L, short=False, Ls=Ls, B=B)
            full = self.stack_fwd("segmem_encoder", x, B, L, self.segmem_num_layers, False, p=0.0, tape=tape)
            return full.view(B, L, d)[:, :Ls]
        pre, b = "segmem_encoder", "segmem_encoder.block.0.layer"
        Wqkv = self.W(f"{pre}.0.qkv")
        _, xn_full, rstd_full = lib.add_rmsnorm_fwd(x, None, self.ln(f"{b}.0.layer_norm.weight"), eps, dt, write_x1=False)
        kv = lib.gemm_nt(xn_full, Wqkv[inner:])                                     # [B*L, 2*inner]
        xs = x.view(B, L, d)[:, :Ls].contiguous().view(B * Ls, d)
        xns = xn_full.view(B, L, d)[:, :Ls].contiguous().view(B * Ls, d)
        q = lib.gemm_nt(xns, Wqkv[:inner])
        o, lse, o_lo = lib.attn_fwd(q, kv[:, :inner], kv[:, inner:], B, H, Ls, L, False, want_lse=tape is not None,
                                    want_lo=tape is not None)
        y = lib.gemm_nt(o, self.W(f"{pre}.0.o"), out_dtype=self.y_dtype)
        x1, xn1, rstd1 = lib.add_rmsnorm_fwd(xs, y, self.ln(f"{b}.1.layer_norm.weight"), eps, dt)
        if xn1.dtype == torch.bfloat16:
            h, g = lib.gemm_nt_geglu(xn1, self.W(f"{pre}.0.wi"))
        else:
            h = lib.gemm_nt(xn1, self.W(f"{pre}.0.wi"))
            g = lib.geglu_fwd(h)
        y2 = lib.gemm_nt(g, self.W(f"{pre}.0.wo"), out_dtype=self.y_dtype)
        x2, out, rstd2 = lib.add_rmsnorm_fwd(x1, y2, self.ln(f"{pre}.final_layer_norm.weight"), eps, dt)
        if tape is not None:
            tape.push(kind="seg_in", ids=ids.reshape(-1), emb=emb_a, L=L, short=True, Ls=Ls, B=B, x=x,
                      xn_full=xn_full, rstd_full=rstd_full, kv=kv, xns=xns, q=q, o=o, o_lo=o_lo, lse=lse, x1=x1, xn1=xn1,
                      rstd1=rstd1, h=h, g=g, x2=x2, rstd2=rstd2)
        return out.view(B, Ls, d)

    def segmem_bwd(self, tape, d_mem):
        """d_mem: fp32 [B, Ls, d] gradient w.r.t. the memory vectors."""
        f, d, H, inner = self.flat, self.d, self.H, self.inner
        if tape.ops[-1]["kind"] != "seg_in":                    # generic multi-layer path
            fin = next(o for o in reversed(tape.ops) if o["kind"] == "seg_in")
            B, L, Ls = fin["B"], fin["L"], fin["Ls"]
            d_full = torch.zeros(B, L, d, device=d_mem.device, dtype=torch.float32)
            d_full[:, :Ls] = d_mem
            dx = self.stack_bwd(tape, d_full.view(B * L, d))
            t = tape.pop()
        else:
            t = tape.pop()
            B, L, Ls = t["B"], t["L"], t["Ls"]
            pre, b = "segmem_encoder", "segmem_encoder.block.0.layer"
            GW, WT = f.GW(f"{pre}.0.qkv"), self.WT(f"{pre}.0.qkv")
            d_out = d_mem.contiguous().view(B * Ls, d)
            dx2, dy2 = self._norm_bwd(d_out, None, t["x2"], t["rstd2"], self.ln(f"{pre}.final_layer_norm.weight"),
                                           f.grad(f"{pre}.final_layer_norm.weight"))
            self.wgrad(dy2, t["g"], f.GW(f"{pre}.0.wo"))
            dg = lib.gemm_nt(dy2, self.WT(f"{pre}.0.wo"))
            dh = lib.geglu_bwd(t["h"], dg)
            self.wgrad(dh, t["xn1"], f.GW(f"{pre}.0.wi"))
            dxn1 = lib.gemm_nt(dh, self.WT(f"{pre}.0.wi"), out_dtype=self.y_dtype)
            dx1, dy = self._norm_bwd(dxn1, dx2, t["x1"], t["rstd1"], self.ln(f"{b}.1.layer_norm.weight"),
                                          f.grad(f"{b}.1.layer_norm.weight"), dx1=dx2)
            self.wgrad(dy, t["o"], f.GW(f"{pre}.0.o"))
            do = lib.gemm_nt(dy, self.WT(f"{pre}.0.o"))
            kv = t["kv"]
            dq, dkv = torch.empty_like(t["q"]), torch.empty_like(kv)
            lib.attn_bwd(t["q"], kv[:, :inner], kv[:, inner:], t["o"], do, t["lse"], dq, dkv[:, :inner],
                         dkv[:, inner:], B, H, Ls, L, False, o_lo=t["o_lo"])
            self.wgrad(dq, t["xns"], GW[:inner])
            self.wgrad(dkv, t["xn_full"], GW[inner:])
            dxn_full = lib.gemm_nt(dkv, WT[:, inner:], out_dtype=torch.float32)          # [B*L, d]
            dxns = lib.gemm_nt(dq, WT[:, :inner], out_dtype=torch.float32)               # [B*Ls, d]
            dxn_full.view(B, L, d)[:, :Ls] += dxns.view(B, Ls, d)
            dres = torch.zeros(B, L, d, device=dx1.device, dtype=torch.float32)
            dres[:, :Ls] = dx1.view(B, Ls, d)
            dx, _ = self._norm_bwd(dxn_full, dres.view(B * L, d), t["x"], t["rstd_full"],
                                        self.ln(f"{b}.0.layer_norm.weight"), f.grad(f"{b}.0.layer_norm.weight"),
                                        want_dy=False)
        assert t["kind"] == "seg_in"
        dsrc = lib.dropmask_cast(dx, out_dtype=self.dt)
        self.wgrad(dsrc, t["emb"], f.GW("segmem_proj"))
        demb = lib.gemm_nt(dsrc, self.WT("segmem_proj"), out_dtype=torch.float32)
        lib.embed_bwd(t["ids"], demb, f.grad("decoder_embed_tokens.weight"), t["L"], shift=False,
                      pad_id=self.cfg["pad_token_id"])

    @staticmethod
    def prev_row_ids(labels, start_id, pad_id):
        """V1/V2 memory ids (models/t5_segmem_v2.py:126-133) from the shifted decoder inputs."""
        B, L = labels.shape
        dec = torch.cat([labels.new_full((B, 1), start_id), labels[:, :-1]], 1)
        dec = dec.masked_fill(dec == -100, pad_id)
        seg = torch.cat([dec[:, 1:], dec.new_zeros(B, 1)], 1)
        # row 0's memory ids: [1, 0, 0, ...] — built on the device (an indexed host write would be a host-to-device copy,
        # which a hipGraph capture of the step does not allow)
        dummy = (torch.arange(L, device=dec.device) == 0).to(dec.dtype).unsqueeze(0)
        return torch.cat([dummy, seg[:-1]], 0).contiguous()

    # ---- full forward / backward --------------------------------------------------------------------------
    def forward(self, mel, labels, targets_prev=None, training=False, need_grad=False, want_logits=True):
        """Returns (logits [B, Ld, V] fp32, tape or None) — or, with want_logits=False, the final-normed decoder states
        [B*Ld, d] in the compute dtype instead of the logits (the trainer feeds them to the fused lm_head + CE call, so
        the 403 MB of f32 logits of a 64-segment batch never exist).  Mirrors get_model_outputs of the four
        reference model classes (models/t5.py:99-180, t5_segmem.py:68-170, t5_segmem_v2.py:64-167,
        t5_segmem_v2_with_prev.py:60-153)."""
        cfg, f = self.cfg, self.flat
        if not mel.is_cuda:
            raise RuntimeError("MR-MT3 MI355X path needs device tensors (no CPU fallback)")
        self.prepare(need_grad)
        p = float(cfg["dropout_rate"]) if training else 0.0
        tape = Tape() if need_grad else None
        B, Le, d = mel.shape
        Ld = labels.shape[1]
        labels = labels.contiguous()
        Ls = self.segmem_length
        enc = self.encode(mel, p=p, tape=tape)                                  # [B*Le, d]
        table = f.master("decoder_embed_tokens.weight")
        start, pad = cfg["decoder_start_token_id"], cfg["pad_token_id"]
        pos = self.pos(mel.device)
        variant = self.variant
        mem = None
        if variant == "segmem_v2_with_prev":
            assert targets_prev is not None
            targets_prev.masked_fill_(targets_prev == -100, pad)                 # in place, like the reference (:119)
        if Ls == 0:
            # model_segmem_length=0 (the reference's no-memory ablation): `[:, :0]` leaves nothing to attend
            # to or prepend, so the step is the plain MT3 one and the memory block receives no gradient
            variant = "t5"
        if variant != "t5":
            if variant == "segmem_v2_with_prev":
                ids = targets_prev.contiguous()
            else:
                ids = self.prev_row_ids(labels, start, pad)
            Lm = ids.shape[1]
            mem = self.segmem(ids, B, Lm, tape=tape)                             # [B, Ls, d]
            Ls = mem.shape[1]
        s_emb = self._sid()
        if variant == "segmem_v1":
            # memory is PREPENDED to the decoder input embeddings (t5_segmem.py:138-160)
            emb = lib.embed_fwd(labels.view(-1), table, None, Ld, shift=True, start_id=start, pad_id=pad)
            xin = torch.cat([mem.float(), emb.view(B, Ld, d)], 1).contiguous()
            Lx = Ld + Ls
            x = lib.addpos_fwd(xin.view(B * Lx, d), pos, Lx, p=p, seed=self.seed, step=self.step_dev, stream_id=s_emb)
            enc_cat, Lc = enc, Le
        else:
            Lx = Ld
            x = lib.embed_fwd(labels.view(-1), table, pos, Ld, shift=True, start_id=start, pad_id=pad, p=p,
                              seed=self.seed, step=self.step_dev, stream_id=s_emb)
            if mem is not None:
                enc_cat = torch.cat([enc.view(B, Le, d), mem], 1).contiguous().view(-1, d)
                Lc = Le + Ls
            else:
                enc_cat, Lc = enc, Le
        if tape is not None:
            tape.push(kind="dec_in", s_emb=s_emb, p=p, labels=labels, B=B, Le=Le, Ld=Ld, Lx=Lx, Lc=Lc, Ls=Ls,
                      Lm=(ids.shape[1] if variant != "t5" else 0), variant=variant)
        head_f32 = self.head_dtype == "f32" and tape is None and self.dt == torch.bfloat16
        dec = self.stack_fwd("decoder", x, B, Lx, cfg["num_decoder_layers"], True, enc=enc_cat, Le=Lc, p=p, tape=tape,
                             out_dtype=torch.float32 if head_f32 else None)
        if variant == "segmem_v1":
            dec = dec.view(B, Lx, d)[:, Ls:].contiguous().view(B * Ld, d)
        if tape is not None:
            tape.push(kind="head", dec=dec)
        if not want_logits:
            return dec, tape
        w_head = f.W("lm_head", torch.float32) if head_f32 else self.W("lm_head")
        logits = lib.gemm_nt(dec, w_head, out_dtype=torch.float32)               # [B*Ld, V]
        return logits.view(B, Ld, self.V), tape

    def backward(self, tape, dlogits, on_layer_done=None):
        """dlogits: [B*Ld, V] in the compute dtype (anything else is cast).  Accumulates into flat.G.  With an fp32 engine
        every product, the attention backward and the row-wise kernels run in exact f32 (the reference's own training
        precision, config/config_slakh_segmem.yaml:47): slow, and the tight pin of this hand-written tape against autograd."""
        f, cfg, d = self.flat, self.cfg, self.d
        variant, Ls = self.variant, self.segmem_length
        t = tape.pop()
        assert t["kind"] == "head"
        dl = dlogits.reshape(-1, self.V)
        if dl.dtype != self.dt:
            dl = self._act(dl)
        self.wgrad(dl, t["dec"], f.GW("lm_head"))
        d_dec = lib.gemm_nt(dl, self.WT("lm_head"), out_dtype=torch.float32)
        if on_layer_done is not None:
            on_layer_done("lm_head", 0)
        # peek the decoder-input record (it sits below the decoder stack's records)
        din = next(o for o in reversed(tape.ops) if o["kind"] == "dec_in")
        B, Le, Ld, Lx, Lc, Ls = din["B"], din["Le"], din["Ld"], din["Lx"], din["Lc"], din["Ls"]
        variant = din["variant"]                                # "t5" when segmem_length is 0
        if variant == "segmem_v1":
            full = torch.zeros(B, Lx, d, device=d_dec.device, dtype=torch.float32)
            full[:, Ls:] = d_dec.view(B, Ld, d)
            d_dec = full.view(B * Lx, d)
        d_enc_cat = torch.empty(B * Lc, d, device=d_dec.device, dtype=torch.float32)      # written by stack_bwd's one d_enc product
        dx = self.stack_bwd(tape, d_dec, d_enc=d_enc_cat, on_layer_done=on_layer_done)
        t = tape.pop()
        assert t["kind"] == "dec_in"
        table_g = f.grad("decoder_embed_tokens.weight")
        start, pad = cfg["decoder_start_token_id"], cfg["pad_token_id"]
        d_mem = None
        if variant == "segmem_v1":
            dxm = lib.dropmask_cast(dx, p=t["p"], seed=self.seed, step=self.step_dev, stream_id=t["s_emb"],
                                    out_dtype=self.dt).float().view(B, Lx, d)
            d_mem = dxm[:, :Ls]
            lib.embed_bwd(t["labels"].view(-1), dxm[:, Ls:].contiguous().view(-1, d), table_g, Ld, shift=True,
                          start_id=start, pad_id=pad)
            d_enc = d_enc_cat
        else:
            lib.embed_bwd(t["labels"].view(-1), dx, table_g, Ld, shift=True, start_id=start, pad_id=pad, p=t["p"],
                          seed=self.seed, step=self.step_dev, stream_id=t["s_emb"])
            if variant != "t5":
                dcat = d_enc_cat.view(B, Lc, d)
                d_enc = dcat[:, :Le].contiguous().view(-1, d)
                d_mem = dcat[:, Le:]
            else:
                d_enc = d_enc_cat
        if d_mem is not None:
            self.segmem_bwd(tape, d_mem)
            if on_layer_done is not None:
                on_layer_done("segmem", 0)                     # the memory encoder's gradients are final (their bucket may leave)
        self.encode_bwd(tape, d_enc, on_layer_done=on_layer_done)
        self.join_wgrad()
        assert not tape.ops, [o["kind"] for o in tape.ops]
