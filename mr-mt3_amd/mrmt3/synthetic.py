"""Synthetic weights and inputs for the MR-MT3 hot path.

No trained checkpoint ships with the reference (`pretrained/mt3.pth` is a git-LFS pointer), so
every parity and throughput run uses the *golden weight recipe*: one frozen
`numpy.random.RandomState` stream per state-dict key, keys taken in sorted order.  The recipe is
pure data generation — it is shared by the product path, the oracle and the fixture generator so
that the same 184 MB of weights can be regenerated on the GPU box instead of being shipped.

Scales follow the T5 initialisation the reference inherits from HF (`T5PreTrainedModel._init_weights`):
q ~ (d_model*d_kv)^-1/2 (attention is unscaled), k/v/wi ~ d_model^-1/2, o ~ (H*d_kv)^-1/2,
wo ~ d_ff^-1/2, embeddings ~ 1; `lm_head` uses d_model^-1/2 so logits are O(1) like a trained model.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np

T5_SMALL = dict(
    d_model=512, d_kv=64, d_ff=1024, num_heads=6, num_layers=8, num_decoder_layers=8,
    vocab_size=1536, dropout_rate=0.1, layer_norm_epsilon=1e-6, pad_token_id=0, eos_token_id=1,
    unk_token_id=2, decoder_start_token_id=0, feed_forward_proj="gated-gelu",
    tie_word_embeddings=False, is_encoder_decoder=True, use_cache=False,
    initializer_factor=1.0, model_type="t5", output_past=True,
    architectures=["T5ForConditionalGeneration"],
)


def state_dict_shapes(cfg: dict, segmem_num_layers: int = 0) -> "OrderedDict[str, tuple]":
    """Canonical (alias-free) parameter names and shapes, reference `models/t5.py:47-77`,
    `models/t5_segmem.py:48-66`.  Aliases (`encoder.embed_tokens.weight` == `proj.weight`, ...)
    are added by the model's `state_dict()`; they carry no extra data."""
    d, dk, H, dff, V = cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"], cfg["vocab_size"]
    inner = dk * H
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["proj.weight"] = (d, d)
    s["decoder_embed_tokens.weight"] = (V, d)

    def stack(prefix, n_layers, is_decoder):
        for i in range(n_layers):
            b = f"{prefix}.block.{i}.layer"
            for n in ("q", "k", "v"):
                s[f"{b}.0.SelfAttention.{n}.weight"] = (inner, d)
            s[f"{b}.0.SelfAttention.o.weight"] = (d, inner)
            s[f"{b}.0.layer_norm.weight"] = (d,)
            ff = 1
            if is_decoder:
                for n in ("q", "k", "v"):
                    s[f"{b}.1.EncDecAttention.{n}.weight"] = (inner, d)
                s[f"{b}.1.EncDecAttention.o.weight"] = (d, inner)
                s[f"{b}.1.layer_norm.weight"] = (d,)
                ff = 2
            s[f"{b}.{ff}.DenseReluDense.wi_0.weight"] = (dff, d)
            s[f"{b}.{ff}.DenseReluDense.wi_1.weight"] = (dff, d)
            s[f"{b}.{ff}.DenseReluDense.wo.weight"] = (d, dff)
            s[f"{b}.{ff}.layer_norm.weight"] = (d,)
        s[f"{prefix}.final_layer_norm.weight"] = (d,)

    stack("encoder", cfg["num_layers"], False)
    stack("decoder", cfg["num_decoder_layers"], True)
    s["lm_head.weight"] = (V, d)
    if segmem_num_layers:
        s["segmem_proj.weight"] = (d, d)
        stack("segmem_encoder", segmem_num_layers, False)
    return s


def _std_for(key: str, cfg: dict) -> float:
    d, dk, H, dff = cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"]
    leaf = key.rsplit(".", 2)[-2]
    if leaf == "q":
        return (d * dk) ** -0.5
    if leaf in ("k", "v", "wi_0", "wi_1"):
        return d ** -0.5
    if leaf == "o":
        return (H * dk) ** -0.5
    if leaf == "wo":
        return dff ** -0.5
    if leaf in ("proj", "segmem_proj", "lm_head"):
        return d ** -0.5
    if leaf == "decoder_embed_tokens":
        return 1.0
    raise KeyError(key)


def golden_weights(cfg: dict = T5_SMALL, segmem_num_layers: int = 0, seed: int = 365,
                   keys=None) -> "OrderedDict[str, np.ndarray]":
    """fp32 weights.  Stream for key k = RandomState(crc32(k) ^ seed): independent of which other
    keys exist, so T5 and segmem models share their common tensors exactly."""
    shapes = state_dict_shapes(cfg, segmem_num_layers)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k in sorted(shapes) if keys is None else keys:
        shp = shapes[k]
        rs = np.random.RandomState((zlib.crc32(k.encode()) ^ seed) & 0x7FFFFFFF)
        if len(shp) == 1:  # T5LayerNorm scale
            w = 1.0 + 0.1 * rs.standard_normal(shp)
        else:
            w = _std_for(k, cfg) * rs.standard_normal(shp)
        out[k] = w.astype(np.float32)
    return OrderedDict((k, out[k]) for k in shapes if k in out)


def synth_audio(batch: int, n_samples: int = 32768, seed: int = 365) -> np.ndarray:
    """Uniform [-1,1) fp32 audio, SURVEY §8d."""
    rs = np.random.RandomState(seed)
    return (rs.uniform(-1.0, 1.0, size=(batch, n_samples))).astype(np.float32)


def synth_mel(batch: int, frames: int = 256, bins: int = 512, seed: int = 365) -> np.ndarray:
    rs = np.random.RandomState(seed + 1)
    return rs.uniform(0.0, 1.0, size=(batch, frames, bins)).astype(np.float32)


def synth_labels(batch: int, length: int = 1024, seed: int = 365, full: bool = True,
                 mean_len: int = 300) -> np.ndarray:
    """Event ids uniform in [3,1391).  `full`: EOS(1) at the last position, no padding (throughput
    worst case).  Otherwise "Slakh-shaped": geometric length, EOS, then -100 padding
    (reference `dataset/dataset_2_random.py:292-306`)."""
    rs = np.random.RandomState(seed + 2)
    lab = rs.randint(3, 1391, size=(batch, length)).astype(np.int64)
    if full:
        lab[:, -1] = 1
        return lab
    for b in range(batch):
        n = int(min(length - 1, max(1, rs.geometric(1.0 / mean_len))))
        lab[b, n] = 1
        lab[b, n + 1:] = -100
    return lab


def bench_shape_inputs(batch: int = 16, length: int = 1024, seed: int = 4242):
    """Inputs of the bench-shaped parity tests (decoder rows = batch * 1024 >= 16384, encoder rows >= 4096, i.e. the
    shapes at which the tall-GEMM, fused wi+GEGLU and grouped weight-gradient kernels dispatch): mel, labels with even
    rows full-length and odd rows Slakh-shaped (EOS then -100 padding), and an independent targets_prev stream."""
    mel = synth_mel(batch, seed=seed)
    lab = synth_labels(batch, length, seed=seed + 1, full=False, mean_len=500)
    lab[::2] = synth_labels(batch, length, seed=seed + 2, full=True)[::2]
    prev = synth_labels(batch, length, seed=seed + 3, full=False, mean_len=500)
    return mel, lab, prev


def long_shape_inputs(batch: int = 2, frames: int = 2048, length: int = 1024, seed: int = 5151):
    """Inputs of the long-context parity test (BASELINE configs[4], config_slakh_segmem_finetune.yaml with mel_length 2048):
    `batch` segments of `frames` mel frames, 1024-token targets — row 0 full length, the others Slakh-shaped (EOS, then
    -100 padding) — and an independent targets_prev stream."""
    mel = synth_mel(batch * (frames // 256), seed=seed).reshape(batch, frames, 512)
    lab = synth_labels(batch, length, seed=seed + 1, full=False, mean_len=500)
    lab[0] = synth_labels(batch, length, seed=seed + 2, full=True)[0]
    prev = synth_labels(batch, length, seed=seed + 3, full=False, mean_len=500)
    return mel, lab, prev


def sinusoid_table(n_pos: int, dim: int):
    """`FixedPositionalEmbedding`, reference `models/t5.py:705-719`: [sin | cos] halves (not
    interleaved).  Computed with torch CPU fp32 ops in the reference's own op order so the table
    is bit-identical to the one the reference builds on every call; returns a torch tensor."""
    import torch
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(n_pos).type_as(inv_freq)
    ang = torch.einsum("i , j -> i j", t, inv_freq)
    return torch.cat((ang.sin(), ang.cos()), dim=-1)
