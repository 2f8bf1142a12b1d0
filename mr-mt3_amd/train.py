"""MI355X training driver — the build's counterpart of the reference's `train.py:21-116`.

Consumes the reference's Hydra YAML unchanged (through hydra when installed, else
`mrmt3.hydra_lite`), instantiates the task named by `cfg.model._target_`, and runs
`mrmt3.trainer.Trainer` (fused CE, hand-written backward, RCCL gradient exchange, one-launch AdamW),
one process per GPU:

    python train.py --config-dir /path/to/reference/config --config-name config_slakh_segmem \
        model=MT3NetSegMemV2WithPrev dataset=SlakhPrevAugment +synthetic=True +max_steps=100
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...      # data parallel

Data.  With `+synthetic=True` the loop runs `+max_steps` (default 10) synthetic Slakh-shaped batches (raw audio,
log-mel on the GPU).  Without it the configured `cfg.dataset.train` / `cfg.dataset.val` are instantiated and wrapped
in `torch.utils.data.DataLoader(**cfg.dataloader.*, collate_fn=cfg.dataset.collate_fn)` exactly like
`train.py:48-59`: batches are the reference's `(inputs mel, targets[, targets_prev])` tuples, the run lasts
`num_epochs` epochs (or `+max_steps`), and `val_loss` is evaluated every `trainer.check_val_every_n_epoch` epochs.
The Slakh/ComMU dataset classes themselves are the reference's `dataset/` package (librosa, note_seq, ...: out of
scope): put it on PYTHONPATH.  If it cannot be imported the driver STOPS with that message — it
never silently substitutes synthetic data.

`cfg.path` has the reference's meaning (`train.py:61-92`): a `.ckpt` resumes weights, AdamW moments and
the step counter; a `.pth` only loads weights (`strict=False`); anything else is an error.  At the end
rank 0 writes `<output_dir>/<model_type>_<dataset_type>/version_0/checkpoints/last.ckpt` (Lightning
layout) and `last.pt` (bare state dict, no `model.` prefix) like `train.py:105-116`; `+output_dir=...`
picks the root (default: the working directory, where Hydra would have put it).
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from mrmt3 import hydra_lite  # noqa: E402


def synthetic_batches(cfg, rank, device, steps, with_prev):
    from mrmt3.synthetic import synth_audio, synth_labels
    B = int(cfg.dataloader.train.batch_size) * int(cfg.num_rows_per_batch)
    n = int(cfg.mel_length) * 128
    for it in range(steps):
        seed = 365 + 1000 * rank + it
        audio = torch.from_numpy(synth_audio(B, n, seed=seed)).to(device)
        labels = torch.from_numpy(synth_labels(B, int(cfg.event_length), seed=seed, full=False)).to(device)
        prev = torch.from_numpy(synth_labels(B, int(cfg.event_length), seed=seed + 7, full=False)).to(device) if with_prev else None
        yield audio, labels, prev


def real_loaders(cfg):
    """DataLoaders over the configured datasets (train.py:48-59).  Raises with a clear message when the dataset
    package (the reference's `dataset/`, not part of this path) cannot be imported."""
    from torch.utils.data import DataLoader
    try:
        train_set = hydra_lite.instantiate(cfg.dataset.train)
        val_set = hydra_lite.instantiate(cfg.dataset.val)
        collate = hydra_lite.get_method(str(cfg.dataset.collate_fn))
    except ImportError as e:
        raise RuntimeError(
            f"cannot instantiate cfg.dataset ({cfg.dataset.train.get('_target_')}): {e}.  The dataset classes are the "
            "reference's `dataset/` package (needs librosa / note_seq): put it on PYTHONPATH, or pass +synthetic=True "
            "to run synthetic Slakh-shaped batches.") from e
    kw_t = {k: v for k, v in dict(cfg.dataloader.train).items()}
    kw_v = {k: v for k, v in dict(cfg.dataloader.val).items()}
    return (DataLoader(train_set, collate_fn=collate, **kw_t), DataLoader(val_set, collate_fn=collate, **kw_v))


def loader_batches(loader, device, epochs, max_steps):
    """(epoch, inputs, targets, targets_prev) from the reference's collated batches, moved to the device."""
    n = 0
    for ep in range(epochs):
        for batch in loader:
            if max_steps is not None and n >= max_steps:
                return
            inputs, targets = batch[0], batch[1]
            prev = batch[2] if len(batch) > 2 else None
            yield ep, inputs.to(device, non_blocking=True), targets.to(device, non_blocking=True), \
                (None if prev is None else prev.to(device, non_blocking=True))
            n += 1


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", "--config-path", dest="config_dir", required=True)   # both spellings appear in the reference's scripts
    ap.add_argument("--config-name", default="config")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args(argv)
    cfg = hydra_lite.compose(a.config_dir, a.config_name, a.overrides)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.manual_seed(int(cfg.seed))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    task = hydra_lite.instantiate(cfg.model, optim_cfg=cfg.optim, eval_cfg=cfg.get("eval"))
    assert cfg.model_type == cfg.model._target_.split(".")[-1], "model_type and model target mismatched"   # train.py:36
    task.to(device)
    from mrmt3.trainer import Trainer
    from utils import cosine_warmup_lambda
    finetune = type(task).__name__.endswith("FineTune")
    lam = None if finetune else cosine_warmup_lambda(int(cfg.optim.warmup_steps),
                                                     int(cfg.optim.num_steps_per_epoch) * int(cfg.optim.num_epochs),
                                                     min_lr=float(cfg.optim.min_lr))
    trainer = Trainer(task.model, lr=float(cfg.optim.lr), lr_lambda=lam,
                      weighted_loss=type(task).__name__ == "MT3NetWeightedLoss")
    with_prev = "WithPrev" in type(task).__name__
    synthetic = bool(cfg.get("synthetic", False))
    max_steps = cfg.get("max_steps")
    max_steps = None if max_steps is None else int(max_steps)
    train_loader = val_loader = None
    if not synthetic:
        train_loader, val_loader = real_loaders(cfg)
    start = 0
    path = cfg.get("path")
    if path is not None and str(path) != "":
        path = str(path)
        if path.endswith(".ckpt"):
            start = trainer.resume(path)
            if rank == 0:
                print(f"Resuming from {path} at step {start}", flush=True)
        elif path.endswith(".pth"):
            if rank == 0:
                print(f"Loading weights from {path}...", flush=True)
            trainer.resume(path)
        else:
            raise ValueError(f"Invalid extension for path: {path}")
    log_every = max(1, int(cfg.trainer.get("log_every_n_steps", 100)))
    if synthetic:
        steps = 10 if max_steps is None else max_steps
        for it, (audio, labels, prev) in enumerate(synthetic_batches(cfg, rank, device, steps, with_prev), start):
            loss = trainer.train_step(audio, labels, prev, audio=True)
            if rank == 0 and (it % log_every == 0 or it == start + steps - 1):
                print(f"step {it} train_loss {loss.item():.4f}", flush=True)
    else:
        val_every = max(1, int(cfg.trainer.get("check_val_every_n_epoch", 1)))
        it, last_ep, loss = start, 0, None

        def validate(ep):
            tot, n = 0.0, 0
            for b in val_loader:
                prev = b[2].to(device) if len(b) > 2 else None
                tot += float(trainer.eval_loss(b[0].to(device), b[1].to(device), prev).item())
                n += 1
            if rank == 0 and n:
                print(f"epoch {ep} val_loss {tot / n:.4f}", flush=True)

        for ep, mel, labels, prev in loader_batches(train_loader, device, int(cfg.num_epochs), max_steps):
            if ep != last_ep and ep % val_every == 0:
                validate(last_ep)
            last_ep = ep
            loss = trainer.train_step(mel, labels, prev, audio=False)
            if rank == 0 and it % log_every == 0:
                print(f"step {it} train_loss {loss.item():.4f}", flush=True)
            it += 1
        validate(last_ep)
    if rank == 0:
        out_dir = os.path.join(str(cfg.get("output_dir", ".")), f"{cfg.model_type}_{cfg.dataset_type}",
                               "version_0", "checkpoints")
        os.makedirs(out_dir, exist_ok=True)
        trainer.save_checkpoint(os.path.join(out_dir, "last.ckpt"))
        trainer.save_checkpoint(os.path.join(out_dir, "last.pt"))
        print(f"Saved model in {os.path.join(out_dir, 'last.pt')}.", flush=True)
    if world > 1:
        dist.destroy_process_group()
    return task


if __name__ == "__main__":
    main()
