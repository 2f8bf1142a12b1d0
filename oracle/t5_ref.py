"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the MR-MT3 model path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
It is a plain-PyTorch fp32 restatement (no HuggingFace import, no reference import) of:

  * `models/t5.py:47-77`     parameter set (M1)           -> state dict consumed as-is
  * `models/t5.py:99-180`    get_model_outputs (M2/M3)     -> `forward_logits`
  * `models/t5.py:251-302`   batched greedy generate (M4)  -> `generate_t5`
  * `models/t5.py:478-702`   T5Stack.forward (M5)          -> `t5_stack`
  * `models/t5.py:705-719`   FixedPositionalEmbedding (M6) -> `pos_emb`
  * HF transformers==4.18.0 `modeling_t5.py` T5Block / T5LayerNorm / T5Attention /
    T5DenseGatedGeluDense (M7; third-party, not vendored in /root/reference; arithmetic restated
    from SURVEY.md §3.2 and checked against the installed 5.15.0 source) -> `rms_norm`, `attention`,
    `ff_gated_gelu`
  * HF `_shift_right`, 4.18 `get_extended_attention_mask` additive -10000 causal mask (M8)
  * `models/t5_segmem.py:68-170` (S2), `models/t5_segmem_v2.py:64-233` (S3),
    `models/t5_segmem_v2_with_prev.py:60-296` (S4, S5)
  * `tasks/mt3_net.py:32-35` CE loss (T2), `:75-165` weighted loss (T4), `utils.py:53-61` (T3)

PARITY PINNING: the reference holds no tests or golden vectors for this path.  This
oracle is pinned against the reference ITSELF, imported in the build container through the shim in
`tests/golden/make_golden.py`; that script writes `tests/golden/*.npz`, and
`tests/test_oracle_golden.py` re-checks this file against those vectors on every run.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# building blocks (HF 4.18 T5 arithmetic)
# ----------------------------------------------------------------------------------------------

def rms_norm(x, w, eps=1e-6):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * w; no mean subtraction, no bias."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def attention(xq, xkv, wq, wk, wv, wo, n_heads, mask=None):
    """T5Attention without relative bias (reference builds every block with
    has_relative_attention_bias=False, `models/t5.py:487-490`): scores are UNSCALED q.k^T plus an
    additive mask, softmax in fp32."""
    B, Lq, _ = xq.shape
    Lk = xkv.shape[1]
    dk = wq.shape[0] // n_heads
    q = (xq @ wq.t()).view(B, Lq, n_heads, dk).transpose(1, 2)
    k = (xkv @ wk.t()).view(B, Lk, n_heads, dk).transpose(1, 2)
    v = (xkv @ wv.t()).view(B, Lk, n_heads, dk).transpose(1, 2)
    scores = q @ k.transpose(2, 3)
    if mask is not None:
        scores = scores + mask
    p = F.softmax(scores.float(), dim=-1).type_as(scores)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, n_heads * dk)
    return o @ wo.t()


def ff_gated_gelu(x, wi0, wi1, wo):
    return (gelu_new(x @ wi0.t()) * (x @ wi1.t())) @ wo.t()


def pos_emb(seq, dim, offset=0, max_length=5000):
    """`FixedPositionalEmbedding.forward`, `models/t5.py:712-719` (same op order, fp32)."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(max_length).type_as(inv_freq)
    s = torch.einsum("i , j -> i j", t, inv_freq)
    emb = torch.cat((s.sin(), s.cos()), dim=-1)
    return emb[None, offset:offset + seq, :]


def causal_additive_mask(L):
    """4.18 `get_extended_attention_mask` for a decoder with an all-ones attention mask:
    (1 - tril) * -10000.0, shape [1,1,L,L]."""
    ids = torch.arange(L)
    causal = (ids[None, :] <= ids[:, None]).float()
    return ((1.0 - causal) * -10000.0)[None, None]


def shift_right(labels, start_id=0, pad_id=0):
    """HF `_shift_right`: prepend decoder_start_token_id, drop last, replace -100 by pad."""
    out = labels.new_zeros(labels.shape)
    out[..., 1:] = labels[..., :-1].clone()
    out[..., 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def t5_stack(sd, prefix, cfg, x, is_decoder, enc=None, n_layers=None):
    """`T5Stack.forward` (`models/t5.py:507-702`), eval mode (dropout = identity), all-ones masks."""
    H, eps = cfg["num_heads"], cfg["layer_norm_epsilon"]
    L = x.shape[1]
    x = x + pos_emb(L, x.shape[-1])
    mask = causal_additive_mask(L) if is_decoder else None
    if n_layers is None:
        n_layers = cfg["num_decoder_layers"] if is_decoder else cfg["num_layers"]
    for i in range(n_layers):
        b = f"{prefix}.block.{i}.layer"
        g = lambda n: sd[f"{b}.{n}.weight"]
        xn = rms_norm(x, g("0.layer_norm"), eps)
        x = x + attention(xn, xn, g("0.SelfAttention.q"), g("0.SelfAttention.k"),
                          g("0.SelfAttention.v"), g("0.SelfAttention.o"), H, mask)
        ff = 1
        if is_decoder and enc is not None:
            xn = rms_norm(x, g("1.layer_norm"), eps)
            x = x + attention(xn, enc, g("1.EncDecAttention.q"), g("1.EncDecAttention.k"),
                              g("1.EncDecAttention.v"), g("1.EncDecAttention.o"), H, None)
            ff = 2
        xn = rms_norm(x, g(f"{ff}.layer_norm"), eps)
        x = x + ff_gated_gelu(xn, g(f"{ff}.DenseReluDense.wi_0"), g(f"{ff}.DenseReluDense.wi_1"),
                              g(f"{ff}.DenseReluDense.wo"))
    return rms_norm(x, sd[f"{prefix}.final_layer_norm.weight"], eps)


# ----------------------------------------------------------------------------------------------
# model variants
# ----------------------------------------------------------------------------------------------

def encode(sd, cfg, mel):
    return t5_stack(sd, "encoder", cfg, mel @ sd["proj.weight"].t(), False)


def decode_logits(sd, cfg, dec_ids, enc, dec_embeds=None):
    x = sd["decoder_embed_tokens.weight"][dec_ids] if dec_embeds is None else dec_embeds
    y = t5_stack(sd, "decoder", cfg, x, True, enc=enc)
    return y


def segmem_memory(sd, cfg, ids, segmem_length, n_layers=1):
    """embed -> segmem_proj -> 1-layer bidirectional segmem_encoder over ALL positions -> first segmem_length
    (`models/t5_segmem_v2_with_prev.py:121-123`)."""
    emb = sd["decoder_embed_tokens.weight"][ids]
    # `self.segmem_encoder(segmem_embeds)` passes the embeddings POSITIONALLY, i.e. as
    # T5Stack.forward's first parameter `input_ids` (`models/t5.py:507-509`), so the stack applies
    # its `embed_tokens` = `segmem_proj` (a bias-free Linear, `models/t5_segmem.py:56,65`) to them
    # at `models/t5.py:539-540`.  (SURVEY.md §8a row S1 says segmem_proj is never applied; the
    # reference's recorded outputs show it is.)
    emb = emb @ sd["segmem_proj.weight"].t()
    return t5_stack(sd, "segmem_encoder", cfg, emb, False, n_layers=n_layers)[:, :segmem_length]


def _prev_row_ids(dec_ids):
    """V1/V2 memory ids (`models/t5_segmem_v2.py:126-133`): row b gets row b-1's decoder inputs
    shifted left by one with a trailing 0; row 0 gets the dummy [1,0,0,...]."""
    B, L = dec_ids.shape
    seg = torch.cat([dec_ids[:, 1:], dec_ids.new_zeros(B, 1)], dim=1)
    dummy = dec_ids.new_zeros(1, L)
    dummy[0, 0] = 1
    return torch.cat([dummy, seg[:-1]], dim=0).long()


def forward_logits(sd, cfg, mel, labels, variant="t5", targets_prev=None, segmem_length=64):
    """Returns lm_logits [B,Ld,V] for variant in {t5, segmem_v1, segmem_v2, segmem_v2_with_prev}."""
    enc = encode(sd, cfg, mel)
    dec_ids = shift_right(labels, cfg["decoder_start_token_id"], cfg["pad_token_id"])
    if variant == "t5":
        y = decode_logits(sd, cfg, dec_ids, enc)
    elif variant == "segmem_v1":
        mem = segmem_memory(sd, cfg, _prev_row_ids(dec_ids), segmem_length)
        emb = torch.cat([mem, sd["decoder_embed_tokens.weight"][dec_ids]], dim=1)
        y = decode_logits(sd, cfg, None, enc, dec_embeds=emb)[:, segmem_length:]
    elif variant == "segmem_v2":
        mem = segmem_memory(sd, cfg, _prev_row_ids(dec_ids), segmem_length)
        y = decode_logits(sd, cfg, dec_ids, torch.cat([enc, mem], dim=1))
    elif variant == "segmem_v2_with_prev":
        tp = targets_prev.masked_fill(targets_prev == -100, cfg["pad_token_id"])
        mem = segmem_memory(sd, cfg, tp, segmem_length)
        y = decode_logits(sd, cfg, dec_ids, torch.cat([enc, mem], dim=1))
    else:
        raise ValueError(variant)
    return y @ sd["lm_head.weight"].t()


def ce_loss(logits, targets):
    """`tasks/mt3_net.py:32-35`."""
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1), ignore_index=-100)


def weighted_ce_loss(logits, targets, lo=1135, hi=1262, pad_id=-100):
    """`MT3NetWeightedLoss.training_step`, `tasks/mt3_net.py:75-165`: instrument (program) tokens
    are added with weight 2 on top of the plain sum; normaliser = n_instrument + n_nonpad."""
    l = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1), ignore_index=-100,
                        reduction="none")
    t = targets.reshape(-1)
    inst = ((t >= lo) & (t <= hi)).float()
    nonpad = (t != pad_id).float()
    return ((l * nonpad).sum() + 2.0 * (l * inst).sum()) / (inst.sum() + nonpad.sum())


def generate_t5(sd, cfg, mel, max_length=1024, return_margins=False):
    """Reference algorithm of `models/t5.py:251-302`: no KV cache, full prefix recompute."""
    B = mel.shape[0]
    enc = encode(sd, cfg, mel)
    ids = torch.full((B, 1), cfg["decoder_start_token_id"], dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    margins = []
    for _ in range(max_length):
        y = decode_logits(sd, cfg, ids, enc)
        logits = y[:, -1] @ sd["lm_head.weight"].t()
        if return_margins:
            top2 = logits.topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
        nxt = logits.argmax(-1)
        nxt = nxt * unfinished + cfg["pad_token_id"] * (1 - unfinished)
        unfinished = unfinished * (nxt != cfg["eos_token_id"]).long()
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        if unfinished.max() == 0:
            break
    if return_margins:
        return ids, torch.stack(margins, dim=1)
    return ids


def generate_t5_cached(sd, cfg, mel, max_length=1024):
    """The same greedy decode with a self-attention KV cache and the cross-attention K/V projected once (what any
    efficient implementation does, the HIP path included).  NOT the reference's algorithm (`models/t5.py:267-295`
    recomputes the whole prefix per token): it exists as the CPU baseline BASELINE.md §2 asks for next to the no-cache
    one, and `tests/test_oracle_golden.py` checks that it returns the tokens `generate_t5` returns."""
    H, eps, d = cfg["num_heads"], cfg["layer_norm_epsilon"], cfg["d_model"]
    L = cfg["num_decoder_layers"]
    B = mel.shape[0]
    enc = encode(sd, cfg, mel)
    pe = pos_emb(max_length + 1, d)[0]
    blk = lambda i, n: sd[f"decoder.block.{i}.layer.{n}.weight"]
    dk = blk(0, "0.SelfAttention.q").shape[0] // H
    heads = lambda t: t.view(B, -1, H, dk).transpose(1, 2)
    ck = [heads(enc @ blk(i, "1.EncDecAttention.k").t()) for i in range(L)]
    cv = [heads(enc @ blk(i, "1.EncDecAttention.v").t()) for i in range(L)]
    sk = [torch.zeros(B, H, max_length + 1, dk) for _ in range(L)]
    sv = [torch.zeros(B, H, max_length + 1, dk) for _ in range(L)]
    ids = torch.full((B, 1), cfg["decoder_start_token_id"], dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)

    def attend(q, k, v, wo):
        p = F.softmax((q @ k.transpose(2, 3)).float(), dim=-1)
        return (p @ v).transpose(1, 2).reshape(B, 1, H * dk) @ wo.t()

    for t in range(max_length):
        x = sd["decoder_embed_tokens.weight"][ids[:, -1:]] + pe[t]
        for i in range(L):
            xn = rms_norm(x, blk(i, "0.layer_norm"), eps)
            sk[i][:, :, t] = heads(xn @ blk(i, "0.SelfAttention.k").t())[:, :, 0]
            sv[i][:, :, t] = heads(xn @ blk(i, "0.SelfAttention.v").t())[:, :, 0]
            x = x + attend(heads(xn @ blk(i, "0.SelfAttention.q").t()), sk[i][:, :, :t + 1], sv[i][:, :, :t + 1],
                           blk(i, "0.SelfAttention.o"))
            xn = rms_norm(x, blk(i, "1.layer_norm"), eps)
            x = x + attend(heads(xn @ blk(i, "1.EncDecAttention.q").t()), ck[i], cv[i], blk(i, "1.EncDecAttention.o"))
            xn = rms_norm(x, blk(i, "2.layer_norm"), eps)
            x = x + ff_gated_gelu(xn, blk(i, "2.DenseReluDense.wi_0"), blk(i, "2.DenseReluDense.wi_1"),
                                  blk(i, "2.DenseReluDense.wo"))
        logits = rms_norm(x, sd["decoder.final_layer_norm.weight"], eps)[:, 0] @ sd["lm_head.weight"].t()
        nxt = logits.argmax(-1)
        nxt = nxt * unfinished + cfg["pad_token_id"] * (1 - unfinished)
        unfinished = unfinished * (nxt != cfg["eos_token_id"]).long()
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        if unfinished.max() == 0:
            break
    return ids


def generate_segmem_v2(sd, cfg, mel, max_length=1024, segmem_length=64, with_prev=True,
                       return_margins=False):
    """`T5SegMemV2WithPrev.generate` (`models/t5_segmem_v2_with_prev.py:226-296`; with_prev=False
    gives `T5SegMemV2.generate`, `models/t5_segmem_v2.py:169-233`, whose only difference is the
    dummy first-segment memory [1,0,...] instead of [1134,1,0,...])."""
    enc = encode(sd, cfg, mel)
    outs, margins = [], []
    seg_ids = torch.zeros(1, max_length, dtype=torch.long)
    if with_prev:
        seg_ids[0, 0], seg_ids[0, 1] = 1134, 1
    else:
        seg_ids[0, 0] = 1
    for i in range(enc.shape[0]):
        mem = segmem_memory(sd, cfg, seg_ids, segmem_length)
        cur = torch.cat([enc[i:i + 1], mem], dim=1)
        toks = torch.zeros(1, 1, dtype=torch.long)
        m = []
        for _ in range(max_length):
            y = decode_logits(sd, cfg, toks, cur)
            logits = y[:, -1] @ sd["lm_head.weight"].t()
            if return_margins:
                top2 = logits.topk(2, dim=-1).values
                m.append(float(top2[0, 0] - top2[0, 1]))
            nxt = logits.argmax(-1)
            toks = torch.cat([toks, nxt[:, None]], dim=1)
            if int(nxt) == cfg["eos_token_id"]:
                break
        # F.pad with a negative amount truncates: result is exactly max_length long (:287-291)
        toks = F.pad(toks, (0, max_length - toks.shape[1]), value=0)
        outs.append(toks)
        margins.append(m)
        seg_ids = toks
    out = torch.cat(outs, dim=0)
    return (out, margins) if return_margins else out


def generate_segmem_v1(sd, cfg, mel, max_length=1024, segmem_length=64, return_margins=False):
    """`T5SegMem.generate_2` (`models/t5_segmem.py:172-252`): per segment, the memory (previous segment's
    tokens through segmem_proj + segmem_encoder, first `segmem_length` positions) is PREPENDED to the
    decoder input embeddings; the decoder runs causally over [memory ; tokens] and the memory positions
    are sliced off before lm_head.  First segment's memory ids are [1,0,0,...] (`:190-196`)."""
    enc = encode(sd, cfg, mel)
    table = sd["decoder_embed_tokens.weight"]
    outs, margins = [], []
    seg_ids = torch.zeros(1, max_length, dtype=torch.long)
    seg_ids[0, 0] = 1
    for i in range(enc.shape[0]):
        mem = segmem_memory(sd, cfg, seg_ids, segmem_length)
        Ls = mem.shape[1]
        toks = torch.zeros(1, 1, dtype=torch.long)
        m = []
        for _ in range(max_length):
            emb = torch.cat([mem, table[toks]], dim=1)
            y = decode_logits(sd, cfg, None, enc[i:i + 1], dec_embeds=emb)[:, Ls:]
            logits = y[:, -1] @ sd["lm_head.weight"].t()
            if return_margins:
                top2 = logits.topk(2, dim=-1).values
                m.append(float(top2[0, 0] - top2[0, 1]))
            nxt = logits.argmax(-1)
            toks = torch.cat([toks, nxt[:, None]], dim=1)
            if int(nxt) == cfg["eos_token_id"]:
                break
        toks = F.pad(toks, (0, max_length - toks.shape[1]), value=0)     # negative pad truncates (:240-244)
        outs.append(toks)
        margins.append(m)
        seg_ids = toks
    out = torch.cat(outs, dim=0)
    return (out, margins) if return_margins else out


def cosine_lambda(step, num_warmup_steps, num_training_steps, num_cycles=0.5, min_lr=2e-5):
    """`utils.py:53-61`: note `min_lr` floors the LambdaLR *multiplier*."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(min_lr, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
