"""Greedy generation on the MI355X decoder (csrc/decode.hip) with the reference's output contract.

  * `T5ForConditionalGeneration.generate` (models/t5.py:251-302): batched; returns int64
    [B, 1 + steps] starting with token 0, finished rows padded with 0, stops when every row has
    emitted EOS.
  * `T5SegMemV2.generate` / `T5SegMemV2WithPrev.generate` (t5_segmem_v2.py:169-233,
    t5_segmem_v2_with_prev.py:226-296): segments decoded one after the other, each conditioned on
    the previous segment's tokens through the segment-memory encoder; returns [n_seg, max_length].
The encoder, the segment-memory encoder and the cross-attention K/V projections run through the
same engine kernels as training; only the token loop uses the KV-cached step graph.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib


MAX_DECODE_BATCH = 256     # DEC_MAXB of csrc/decode.hip: sequences decoded together (one wave per row x sequence)


class _Weights(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("pos", C.c_void_p), ("lm_head", C.c_void_p), ("final_ln", C.c_void_p)] + \
               [(n, C.POINTER(C.c_void_p)) for n in ("ln_self", "w_qkv", "w_o_self", "ln_cross", "w_q_cross",
                                                     "w_o_cross", "ln_ff", "w_wi", "w_wo")]


class Decoder:
    """Owns one mrmt3_decoder handle (KV cache, scratch, captured graph) for a model."""

    def __init__(self, model, max_batch: int, max_len: int, max_enc_len: int):
        self.model = model
        eng, cfg = model.engine, model.cfg
        self.max_batch, self.max_len, self.max_enc = max_batch, max_len, max_enc_len
        self.dt = eng.dt
        h = C.c_void_p()
        lib._check(lib.load().mrmt3_decoder_create(C.byref(h), cfg["num_decoder_layers"], cfg["d_model"],
                                                   cfg["num_heads"], cfg["d_ff"], cfg["vocab_size"], max_batch,
                                                   max_len, max_enc_len, lib.BF16 if self.dt == torch.bfloat16 else lib.F32,
                                                   cfg["layer_norm_epsilon"]), "decoder_create")
        self.h = h
        self.tokens = torch.zeros(max_batch, max_len + 1, dtype=torch.int64, device=model.device)
        self.pinned = torch.zeros(3, dtype=torch.int32).pin_memory()
        self._wkeep = None
        # hipGraph capture is illegal on the legacy default stream: the token loop runs on its own stream
        self.stream = torch.cuda.Stream(device=model.device)
        # persistent cross-attention K|V buffer: a stable address lets the captured graph be reused
        self.ckv_buf = torch.empty(cfg["num_decoder_layers"] * max_batch * max_enc_len * 2 * eng.inner,
                                   device=model.device, dtype=self.dt)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.load().mrmt3_decoder_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _weights(self):
        m, eng, f = self.model, self.model.engine, self.model.flat
        L = m.cfg["num_decoder_layers"]

        def arr(fn):
            a = (C.c_void_p * L)(*[fn(i).data_ptr() for i in range(L)])
            return a

        b = lambda i: f"decoder.block.{i}.layer"
        keep = dict(
            ln_self=arr(lambda i: f.master(f"{b(i)}.0.layer_norm.weight")),
            w_qkv=arr(lambda i: eng.W(f"decoder.{i}.qkv")),
            w_o_self=arr(lambda i: eng.W(f"decoder.{i}.o")),
            ln_cross=arr(lambda i: f.master(f"{b(i)}.1.layer_norm.weight")),
            w_q_cross=arr(lambda i: eng.W(f"decoder.{i}.cq")),
            w_o_cross=arr(lambda i: eng.W(f"decoder.{i}.co")),
            ln_ff=arr(lambda i: f.master(f"{b(i)}.2.layer_norm.weight")),
            w_wi=arr(lambda i: eng.W(f"decoder.{i}.wi")),
            w_wo=arr(lambda i: eng.W(f"decoder.{i}.wo")),
        )
        w = _Weights()
        w.embed = f.master("decoder_embed_tokens.weight").data_ptr()
        w.pos = eng.pos(m.device).data_ptr()
        w.lm_head = eng.W("lm_head").data_ptr()
        w.final_ln = f.master("decoder.final_layer_norm.weight").data_ptr()
        for k, a in keep.items():
            setattr(w, k, C.cast(a, C.POINTER(C.c_void_p)))
        self._wkeep = (keep, w)
        return w

    def cross_kv(self, enc_cat, B, Lc):
        """[layers][B*Lc][2*inner] K|V projections of the (memory-augmented) encoder states."""
        eng = self.model.engine
        L = self.model.cfg["num_decoder_layers"]
        out = self.ckv_buf[:L * B * Lc * 2 * eng.inner].view(L, B * Lc, 2 * eng.inner)
        for i in range(L):
            lib.gemm_nt(enc_cat, eng.W(f"decoder.{i}.ckv"), out=out[i])
        return out

    def run(self, ckv, B, Lc, max_steps, poll_every=64, prefix=None):
        """Decode up to max_steps tokens for B rows; returns (tokens [B, max_len+1] view, steps run,
        finish_step or -1).  `prefix` [B, n, d] f32: memory rows fed as decoder positions 0..n-1."""
        cfg = self.model.cfg
        l = lib.load()
        w = self._weights()
        self._ckv = ckv
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self._run_on_stream(l, w, ckv, B, Lc, max_steps, poll_every, cfg, prefix)
        cur.wait_stream(self.stream)
        return out

    def _run_on_stream(self, l, w, ckv, B, Lc, max_steps, poll_every, cfg, prefix=None):
        lib._check(l.mrmt3_decoder_begin(self.h, C.byref(w), lib._p(ckv), B, Lc, lib._p(self.tokens),
                                         cfg["decoder_start_token_id"], cfg["eos_token_id"], cfg["pad_token_id"],
                                         lib._stream()), "decoder_begin")
        if prefix is not None:
            n_pre = prefix.shape[1]
            assert prefix.dtype == torch.float32 and prefix.is_contiguous() and prefix.shape[0] == B
            assert n_pre + max_steps <= self.max_len, "prefix + token steps exceed the decoder's max_len"
            self._prefix = prefix        # keep alive while the graph may read it
            lib._check(l.mrmt3_decoder_set_prefix(self.h, lib._p(prefix), n_pre, lib._stream()), "decoder_set_prefix")
            max_steps += n_pre
        done, fin = 0, -1
        while done < max_steps:
            n = min(poll_every, max_steps - done)
            lib._check(l.mrmt3_decoder_run(self.h, n, lib._stream()), "decoder_run")
            done += n
            lib._check(l.mrmt3_decoder_poll(self.h, C.c_void_p(self.pinned.data_ptr()), lib._stream()), "decoder_poll")
            torch.cuda.current_stream().synchronize()
            if int(self.pinned[1]):
                fin = int(self.pinned[2])
                break
        return self.tokens, done, fin

    @property
    def graph_captured(self) -> bool:
        return bool(lib.load().mrmt3_decoder_graph_captured(self.h))


def _decoder_for(model, B, max_len, enc_len) -> Decoder:
    dec = getattr(model, "_decoder", None)
    if dec is None or dec.max_batch < B or dec.max_len < max_len or dec.max_enc < enc_len or \
            dec.dt != model.engine.dt or dec.tokens.device != model.device:
        dec = Decoder(model, max(B, 1), max_len, enc_len)
        model._decoder = dec
    return dec


@torch.no_grad()
def generate(model, inputs, max_length=1024, poll_every=64):
    eng, cfg = model.engine, model.cfg
    if not inputs.is_cuda:
        raise RuntimeError("generate needs device tensors (no CPU fallback)")
    eng.prepare(False)
    B, Le, d = inputs.shape
    enc = eng.encode(inputs.float() if inputs.dtype not in (torch.float32, torch.bfloat16) else inputs)
    if model.VARIANT in ("t5", "segmem_v1"):      # T5SegMem.generate ignores the memory (t5_segmem.py:254-311)
        out = []
        for b0 in range(0, B, MAX_DECODE_BATCH):
            nb = min(MAX_DECODE_BATCH, B - b0)
            dec = _decoder_for(model, nb, max_length, Le)
            ckv = dec.cross_kv(enc.view(B, Le, d)[b0:b0 + nb].reshape(nb * Le, d), nb, Le)
            toks, done, fin = dec.run(ckv, nb, Le, max_length, poll_every)
            steps = (fin + 1) if fin >= 0 else max_length
            out.append((toks[:nb, :steps + 1].clone(), steps))
        if len(out) == 1:
            return out[0][0]
        # the reference stops when ALL rows are finished: pad shorter groups with pad_token_id
        steps = max(s for _, s in out)
        res = torch.full((B, steps + 1), cfg["pad_token_id"], dtype=torch.int64, device=inputs.device)
        r = 0
        for t, s in out:
            res[r:r + t.shape[0], :s + 1] = t
            r += t.shape[0]
        return res
    if model.VARIANT == "segmem_v1":
        raise RuntimeError("T5SegMem.generate is the plain batched decode; memory decode is generate_2")
    # segment-memory models: sequential segments, memory = previous segment's tokens
    Ls = min(model.segmem_length, max_length)            # `[:, :segmem_length]` of a max_length-long sequence
    seg_ids = torch.zeros(1, max_length, dtype=torch.int64, device=inputs.device)
    if model.VARIANT == "segmem_v2_with_prev":
        seg_ids[0, 0], seg_ids[0, 1] = 1134, 1          # tie token + EOS (t5_segmem_v2_with_prev.py:257-258)
    else:
        seg_ids[0, 0] = 1                                # t5_segmem_v2.py:199
    dec = _decoder_for(model, 1, max_length, Le + Ls)
    outs = []
    for i in range(B):
        mem = _memory(eng, seg_ids, 1, max_length, Ls)                     # [1, Ls, d]
        cur = torch.cat([enc.view(B, Le, d)[i:i + 1], mem], 1).contiguous().view(Le + Ls, d)
        ckv = dec.cross_kv(cur, 1, Le + Ls)
        toks, done, fin = dec.run(ckv, 1, Le + Ls, max_length, poll_every)
        steps = (fin + 1) if fin >= 0 else max_length
        row = torch.zeros(1, max_length, dtype=torch.int64, device=inputs.device)
        n = min(steps + 1, max_length)                   # F.pad(..., max_length - len) truncates (:287-291)
        row[0, :n] = toks[0, :n]
        outs.append(row)
        seg_ids = row
    return torch.cat(outs, 0)


def _memory(eng, seg_ids, B, L, Ls):
    """Memory vectors of the previous segment; `segmem_length=0` (the reference's no-memory ablation,
    `[:, :0]`) yields an empty block without touching the memory encoder."""
    if Ls == 0:
        return torch.empty(B, 0, eng.d, device=seg_ids.device, dtype=eng.dt)
    return eng.segmem(seg_ids, B, L)


@torch.no_grad()
def generate_2(model, inputs, max_length=1024, poll_every=64):
    """`T5SegMem.generate_2` (models/t5_segmem.py:172-252): segments one after the other; the previous
    segment's tokens go through the segment-memory encoder and its first `segmem_length` outputs are
    PREPENDED to the decoder's input embeddings.  With the KV cache that is a prefix fill: the memory
    rows are fed as decoder positions 0..Ls-1 (self-attention K/V only), tokens start at position Ls.
    A stable prefix buffer keeps the captured step graph valid across segments."""
    eng, cfg = model.engine, model.cfg
    if not inputs.is_cuda:
        raise RuntimeError("generate_2 needs device tensors (no CPU fallback)")
    Ls = model.segmem_length
    assert max_length >= Ls, "the reference asserts segmem_length memory rows (t5_segmem.py:213)"
    eng.prepare(False)
    B, Le, d = inputs.shape
    enc = eng.encode(inputs)
    seg_ids = torch.zeros(1, max_length, dtype=torch.int64, device=inputs.device)
    seg_ids[0, 0] = 1                                          # t5_segmem.py:190-196
    dec = _decoder_for(model, 1, max_length + Ls, Le)
    pre = getattr(dec, "_prefix_buf", None)
    if pre is None or pre.shape[1] != Ls:
        pre = dec._prefix_buf = torch.empty(1, Ls, d, device=inputs.device, dtype=torch.float32)
    outs = []
    for i in range(B):
        if Ls:
            pre.copy_(eng.segmem(seg_ids, 1, max_length).float().view(1, Ls, d))
        ckv = dec.cross_kv(enc.view(B, Le, d)[i].contiguous(), 1, Le)
        toks, done, fin = dec.run(ckv, 1, Le, max_length, poll_every, prefix=pre if Ls else None)
        steps = (fin + 1) if fin >= 0 else max_length
        row = torch.zeros(1, max_length, dtype=torch.int64, device=inputs.device)
        n = min(steps + 1, max_length)
        row[0, :n] = toks[0, :n]
        outs.append(row)
        seg_ids = row
    return torch.cat(outs, 0)


@torch.no_grad()
def generate_songs(model, songs, max_length=1024, poll_every=64):
    """Several recordings decoded in lockstep with the segment-memory models (V2 / V2WithPrev).

    The reference transcribes one recording at a time because segment i needs segment i-1's tokens
    (t5_segmem_v2_with_prev.py:241-294) — but recordings are independent of each other.  Here row s of the
    decode batch is recording s's CURRENT segment: every recording keeps its own memory chain, and each row
    produces exactly what `generate` produces for that recording alone (same kernels, one wave per row and
    sequence).  `songs`: list of [n_seg_s, Le, 512] device tensors.  Returns a list of [n_seg_s, max_length]
    int64 tensors."""
    eng, cfg = model.engine, model.cfg
    if model.VARIANT not in ("segmem_v2", "segmem_v2_with_prev"):
        raise RuntimeError("generate_songs is for the segment-memory models; plain T5 batches segments directly")
    if not songs:
        return []
    if len(songs) > MAX_DECODE_BATCH:
        out = []
        for i in range(0, len(songs), MAX_DECODE_BATCH):
            out += generate_songs(model, songs[i:i + MAX_DECODE_BATCH], max_length, poll_every)
        return out
    dev = songs[0].device
    if dev.type != "cuda":
        raise RuntimeError("generate_songs needs device tensors (no CPU fallback)")
    eng.prepare(False)
    S = len(songs)
    Le, d = songs[0].shape[1], songs[0].shape[2]
    Ls = min(model.segmem_length, max_length)
    enc = [eng.encode(x).view(x.shape[0], Le, d) for x in songs]           # per recording, all its segments
    first = torch.zeros(max_length, dtype=torch.int64, device=dev)
    if model.VARIANT == "segmem_v2_with_prev":
        first[0], first[1] = 1134, 1
    else:
        first[0] = 1
    prev = [first.clone() for _ in range(S)]
    outs = [[] for _ in range(S)]
    for i in range(max(x.shape[0] for x in songs)):
        live = [s for s in range(S) if i < songs[s].shape[0]]
        B = len(live)
        seg_ids = torch.stack([prev[s] for s in live])                      # [B, max_length]
        mem = _memory(eng, seg_ids, B, max_length, Ls)                     # [B, Ls, d]
        cur = torch.cat([torch.stack([enc[s][i] for s in live]), mem.to(enc[0].dtype)], 1).contiguous()
        dec = _decoder_for(model, B, max_length, Le + Ls)
        ckv = dec.cross_kv(cur.view(B * (Le + Ls), d), B, Le + Ls)
        toks, done, fin = dec.run(ckv, B, Le + Ls, max_length, poll_every)
        rows = toks[:B, :max_length].clone()                               # finished rows are already pad(0)-filled
        if done < max_length:                                              # all rows hit EOS early: the rest is stale
            rows[:, done + 1:] = 0
        for r, s in enumerate(live):
            outs[s].append(rows[r])
            prev[s] = rows[r]
    return [torch.stack(o) for o in outs]
