"""`tools.convert_weight` — T5X (Flax) MT3 checkpoint -> this model's state dict.

Counterpart of the reference's tools/convert_weight.py:7-92, without jax/t5x: the input is the
FLATTENED T5X parameter dict (`target/...` keys -> numpy arrays; `state_utils.flatten_state_dict`
output with the optimizer `state/...` entries dropped, convert_weight.py:95-105).  Flax Dense kernels
are [in, out] and become torch Linear weights [out, in]; scales and the embedding table are copied.
The continuous-input projection fills both `proj.weight` and its alias `encoder.embed_tokens.weight`,
the token embedder both `decoder_embed_tokens.weight` and `decoder.embed_tokens.weight`
(convert_weight.py:78-86).
"""
from __future__ import annotations

import numpy as np
import torch

_ATT = (("q", "query"), ("k", "key"), ("v", "value"), ("o", "out"))
_MLP = ("wi_0", "wi_1", "wo")


def t5x_key_map(config) -> dict:
    """{torch key: (t5x key, transpose?)} for every tensor of the T5 model."""
    m = {}
    for i in range(config["num_layers"]):
        src, dst = f"target/encoder/layers_{i}", f"encoder.block.{i}.layer"
        for t, f in _ATT:
            m[f"{dst}.0.SelfAttention.{t}.weight"] = (f"{src}/attention/{f}/kernel", True)
        m[f"{dst}.0.layer_norm.weight"] = (f"{src}/pre_attention_layer_norm/scale", False)
        for w in _MLP:
            m[f"{dst}.1.DenseReluDense.{w}.weight"] = (f"{src}/mlp/{w}/kernel", True)
        m[f"{dst}.1.layer_norm.weight"] = (f"{src}/pre_mlp_layer_norm/scale", False)
    # the reference loops the decoder over config['num_layers'] too (convert_weight.py:54)
    for i in range(config.get("num_decoder_layers") or config["num_layers"]):
        src, dst = f"target/decoder/layers_{i}", f"decoder.block.{i}.layer"
        for t, f in _ATT:
            m[f"{dst}.0.SelfAttention.{t}.weight"] = (f"{src}/self_attention/{f}/kernel", True)
            m[f"{dst}.1.EncDecAttention.{t}.weight"] = (f"{src}/encoder_decoder_attention/{f}/kernel", True)
        m[f"{dst}.0.layer_norm.weight"] = (f"{src}/pre_self_attention_layer_norm/scale", False)
        m[f"{dst}.1.layer_norm.weight"] = (f"{src}/pre_cross_attention_layer_norm/scale", False)
        for w in _MLP:
            m[f"{dst}.2.DenseReluDense.{w}.weight"] = (f"{src}/mlp/{w}/kernel", True)
        m[f"{dst}.2.layer_norm.weight"] = (f"{src}/pre_mlp_layer_norm/scale", False)
    m["lm_head.weight"] = ("target/decoder/logits_dense/kernel", True)
    m["encoder.final_layer_norm.weight"] = ("target/encoder/encoder_norm/scale", False)
    m["decoder.final_layer_norm.weight"] = ("target/decoder/decoder_norm/scale", False)
    for k in ("decoder.embed_tokens.weight", "decoder_embed_tokens.weight"):
        m[k] = ("target/decoder/token_embedder/embedding", False)
    for k in ("proj.weight", "encoder.embed_tokens.weight"):
        m[k] = ("target/encoder/continuous_inputs_projection/kernel", True)
    return m


def convert_t5x_to_pt(config, flatten_statedict) -> dict:
    """Flattened T5X params -> torch state dict.  Like the reference, entries of the input that the
    map does not consume are passed through unchanged and consumed T5X keys are dropped."""
    config = dict(config) if not isinstance(config, dict) else config
    out = dict(flatten_statedict)
    used = set()
    for dst, (src, transpose) in t5x_key_map(config).items():
        value = np.asarray(flatten_statedict[src])
        out[dst] = torch.from_numpy(np.ascontiguousarray(value.T if transpose else value))
        used.add(src)
    for k in used:
        del out[k]
    assert np.allclose(out["proj.weight"].numpy().T,
                       flatten_statedict["target/encoder/continuous_inputs_projection/kernel"])
    return out


def pt_to_t5x(config, state_dict) -> dict:
    """Inverse map (torch state dict -> flattened T5X params); used to round-trip-test the key map."""
    flat = {}
    for dst, (src, transpose) in t5x_key_map(config).items():
        v = state_dict[dst]
        v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        flat[src] = np.ascontiguousarray(v.T if transpose else v)
    return flat


def main(argv=None):
    """python -m tools.convert_weight flat_params.pkl out.pth  (pickle of the flattened T5X dict)."""
    import argparse
    import json
    import os
    import pickle
    ap = argparse.ArgumentParser()
    ap.add_argument("flat_params")
    ap.add_argument("out")
    ap.add_argument("--config", default=None, help="T5 config json (default: MT3 T5-small)")
    a = ap.parse_args(argv)
    if a.config:
        cfg = json.load(open(a.config))
    else:
        from mrmt3.synthetic import T5_SMALL
        cfg = dict(T5_SMALL)
    flat = pickle.load(open(a.flat_params, "rb"))
    flat = {k: v for k, v in flat.items() if not k.startswith("state")}      # convert_weight.py:100-104
    torch.save(convert_t5x_to_pt(cfg, flat), a.out)
    print("wrote", os.path.abspath(a.out))


if __name__ == "__main__":
    main()
