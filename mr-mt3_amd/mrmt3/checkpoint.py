"""Checkpoint interop: the three on-disk forms the reference reads and writes.

  * bare state dict `.pt` / `.pth` — what `train.py:105-116` exports (`model.` prefix stripped) and
    what `test.py:103-110` / `train.py:76-81` load with `strict=False`;
  * Lightning `.ckpt` — `{"state_dict": {"model.<key>": ...}, "optimizer_states": [...],
    "lr_schedulers": [...], "global_step", "epoch"}` as written by `ModelCheckpoint`
    (`train.py:40`, `config/config.yaml` modelcheckpoint) and consumed by
    `load_from_checkpoint` (`test.py:94-102`) and `trainer.fit(ckpt_path=...)` (`train.py:66-72`);
  * the optimizer state inside a `.ckpt` is torch.optim.AdamW's own layout, indexed by the position
    of each parameter in `model.parameters()`.

This module maps those onto the flat parameter store (one fp32 master buffer + flat AdamW moments)
without going through per-parameter Python loops on the device: every tensor is a view of a flat
buffer.  Pure host/torch code, no kernels.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

LIGHTNING_PREFIX = "model."


def reference_parameter_order(cfg: dict, segmem_num_layers: int = 0) -> list:
    """Names in the order `model.parameters()` yields them in the reference (module registration order of
    models/t5.py:47-77 and models/t5_segmem.py:48-66; shared tensors appear once, under their first name).
    torch.optim state dicts index parameters by this position."""
    names = ["proj.weight", "decoder_embed_tokens.weight"]

    def stack(prefix, n, is_decoder):
        for i in range(n):
            b = f"{prefix}.block.{i}.layer"
            for w in "qkvo":
                names.append(f"{b}.0.SelfAttention.{w}.weight")
            names.append(f"{b}.0.layer_norm.weight")
            ff = 1
            if is_decoder:
                for w in "qkvo":
                    names.append(f"{b}.1.EncDecAttention.{w}.weight")
                names.append(f"{b}.1.layer_norm.weight")
                ff = 2
            for w in ("wi_0", "wi_1", "wo"):
                names.append(f"{b}.{ff}.DenseReluDense.{w}.weight")
            names.append(f"{b}.{ff}.layer_norm.weight")
        names.append(f"{prefix}.final_layer_norm.weight")

    stack("encoder", cfg["num_layers"], False)
    stack("decoder", cfg.get("num_decoder_layers") or cfg["num_layers"], True)
    names.append("lm_head.weight")
    if segmem_num_layers:
        names.append("segmem_proj.weight")
        stack("segmem_encoder", segmem_num_layers, False)
    return names


def strip_prefix(state_dict, prefix: str = LIGHTNING_PREFIX):
    """`train.py:109-115`: keys of the LightningModule's state dict lose their `model.` prefix."""
    return OrderedDict(((k[len(prefix):] if k.startswith(prefix) else k), v) for k, v in state_dict.items())


def read_checkpoint(path: str, map_location="cpu") -> dict:
    """-> {"state_dict" (bare keys), "optimizer" (torch AdamW layout or None), "global_step", "epoch",
    "lr_scheduler" (LambdaLR state or None)} for a `.ckpt`, `.pt` or `.pth` file."""
    if not str(path).endswith((".pt", ".pth", ".ckpt")):
        raise ValueError("Only .pt, .pth, .ckpt files are supported.")          # test.py:85-88
    blob = torch.load(path, map_location=map_location, weights_only=False)
    if isinstance(blob, dict) and "state_dict" in blob and isinstance(blob["state_dict"], dict):
        opt = (blob.get("optimizer_states") or [None])[0]
        sch = (blob.get("lr_schedulers") or [None])[0]
        return dict(state_dict=strip_prefix(blob["state_dict"]), optimizer=opt, lr_scheduler=sch,
                    global_step=int(blob.get("global_step", 0)), epoch=int(blob.get("epoch", 0)),
                    extra=blob.get("mrmt3"))
    return dict(state_dict=strip_prefix(blob), optimizer=None, lr_scheduler=None, global_step=0, epoch=0,
                extra=None)


def load_weights(model, path: str, strict: bool = False):
    """`model.load_state_dict(torch.load(path), strict=False)` for any of the three file forms."""
    ck = read_checkpoint(path)
    return model.load_state_dict(ck["state_dict"], strict=strict)


# ---- AdamW state <-> flat moments ----------------------------------------------------------------------
def adamw_state_from_flat(flat, order, step: int, lr: float, betas, eps: float, weight_decay: float,
                          initial_lr: float | None = None) -> dict:
    """torch.optim.AdamW.state_dict() layout whose exp_avg / exp_avg_sq are views of the flat moments."""
    state = {}
    for i, key in enumerate(order):
        state[i] = {"step": torch.tensor(float(step)), "exp_avg": flat.view(flat.M, key),
                    "exp_avg_sq": flat.view(flat.V, key)}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(order)))}
    if initial_lr is not None:
        group["initial_lr"] = initial_lr
    return {"state": state, "param_groups": [group]}


def adamw_state_to_flat(opt_state: dict, flat, order) -> int:
    """Copy a torch AdamW state dict into the flat moments; returns the step count it was saved at."""
    flat.ensure_adam()
    st = opt_state["state"]
    params = opt_state["param_groups"][0]["params"]
    if len(params) != len(order):
        raise ValueError(f"optimizer state has {len(params)} parameters, the model has {len(order)}")
    step = 0
    with torch.no_grad():
        for pos, key in enumerate(order):
            s = st.get(params[pos])
            if s is None:               # parameter never stepped
                flat.view(flat.M, key).zero_()
                flat.view(flat.V, key).zero_()
                continue
            flat.view(flat.M, key).copy_(s["exp_avg"])
            flat.view(flat.V, key).copy_(s["exp_avg_sq"])
            step = max(step, int(float(s["step"])))
    return step


def lightning_checkpoint(model, trainer=None, epoch: int = 0) -> dict:
    """The dict `ModelCheckpoint` would write for a task wrapping `model` (and, with a trainer, the
    optimizer + scheduler state so that the reference's `trainer.fit(ckpt_path=...)` can resume)."""
    sd = OrderedDict((LIGHTNING_PREFIX + k, v.detach().cpu().clone()) for k, v in model.state_dict().items())
    out = {"epoch": epoch, "global_step": 0, "pytorch-lightning_version": "1.9.0", "state_dict": sd,
           "callbacks": {}, "optimizer_states": [], "lr_schedulers": []}
    # no "loops" key: Lightning's restore_loops() reads state_dict["fit_loop"] whenever the key exists, so an empty
    # dict would raise; without it the loops start fresh and `global_step` comes from the optimizer step counts
    if trainer is not None:
        order = reference_parameter_order(model.cfg, model.segmem_num_layers)
        # torch's LambdaLR holds lambda(N) after N optimizer steps (the lr the NEXT step uses); trainer.lr_dev still
        # holds the lr of the last executed step
        lr_now = (trainer.base_lr * trainer.lr_lambda(trainer.host_step) if trainer.lr_lambda is not None
                  else float(trainer.lr_dev.item()))
        opt = adamw_state_from_flat(model.flat, order, trainer.host_step, lr_now, trainer.betas, trainer.eps,
                                    trainer.wd, initial_lr=trainer.base_lr)
        for s in opt["state"].values():
            s["exp_avg"] = s["exp_avg"].detach().cpu().clone()
            s["exp_avg_sq"] = s["exp_avg_sq"].detach().cpu().clone()
        out["optimizer_states"] = [opt]
        out["global_step"] = trainer.host_step
        # not part of Lightning's layout (ignored by it): lets a resumed run draw the same dropout masks
        out["mrmt3"] = {"dropout_seed": model.engine.seed, "dropout_stream_ctr": model.engine._stream_ctr}
        if trainer.lr_lambda is not None:       # LambdaLR.state_dict() (the lambda itself is not pickled)
            out["lr_schedulers"] = [{"base_lrs": [trainer.base_lr], "last_epoch": trainer.host_step,
                                     "verbose": False, "_step_count": trainer.host_step + 1,
                                     "_get_lr_called_within_step": False, "_last_lr": [lr_now],
                                     "lr_lambdas": [None]}]
    return out
