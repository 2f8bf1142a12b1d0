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


def real_loaders(cfg, world=1, rank=0):
    """DataLoaders over the configured datasets (train.py:48-59).  Raises with a clear message when the dataset
    package (the reference's `dataset/`, not part of this path) cannot be imported.

    With more than one rank each loader gets a `DistributedSampler` — what Lightning's DDP strategy injects into the
    reference's plain DataLoaders (`pl.Trainer(strategy="ddp...")`, config/config.yaml:45): every rank iterates a
    disjoint 1/world shard, an epoch is len(dataset)/world steps (which is what `optim.num_steps_per_epoch` counts)
    and the effective batch is world x the per-rank batch.  Returns (train_loader, val_loader, train_sampler or None)."""
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
    from torch.utils.data.distributed import DistributedSampler
    # The training order is a function of (seed, epoch) at EVERY world size — one rank included, where a plain
    # shuffle=True would draw its permutation from the global RNG and a resumed epoch would come in another order than
    # the interrupted one (samples seen twice, others never: ADVICE r4).  The batches are cut by a sampler that can drop
    # its first k batches at the index level: a resumed epoch does not decode and collate what it then throws away.
    sampler = DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=bool(kw_t.pop("shuffle", False)),
                                 seed=int(cfg.seed))
    batches = SkippingBatchSampler(sampler, int(kw_t.pop("batch_size", 1)), bool(kw_t.pop("drop_last", False)))
    if world > 1:
        kw_v.pop("shuffle", None)
        kw_v["sampler"] = DistributedSampler(val_set, num_replicas=world, rank=rank, shuffle=False)
    return (DataLoader(train_set, collate_fn=collate, batch_sampler=batches, **kw_t),
            DataLoader(val_set, collate_fn=collate, **kw_v), sampler if world > 1 else None)


class SkippingBatchSampler(torch.utils.data.BatchSampler):
    """torch's BatchSampler whose NEXT pass drops its first `skip` batches (index lists, nothing loaded); `len()` stays the
    full epoch (it is the steps-per-epoch of the resume arithmetic).  `set_epoch` goes through to the sampler."""

    def __init__(self, sampler, batch_size, drop_last):
        super().__init__(sampler, batch_size, drop_last)
        self.skip = 0

    def set_epoch(self, epoch):
        if hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(epoch)

    def __iter__(self):
        import itertools
        skip, self.skip = self.skip, 0
        return itertools.islice(super().__iter__(), skip, None)


def resume_position(global_step, steps_per_epoch):
    """(epoch, batches of that epoch already consumed) for a run that has taken `global_step` optimizer steps.  The
    position is derived from the step count alone: a checkpoint's `epoch` field is the epoch in progress OR the one
    just completed depending on when it was written, the step count is unambiguous."""
    if steps_per_epoch <= 0:
        return 0, 0
    return global_step // steps_per_epoch, global_step % steps_per_epoch


def loader_batches(loader, device, epochs, max_steps, start_epoch=0, start_step=0, sampler=None, skip=0):
    """(epoch, inputs, targets, targets_prev) from the reference's collated batches, moved to the device.  A resumed
    run continues at (start_epoch, start_step) and drops the first `skip` batches of that epoch (the ones the
    interrupted run consumed): `max_steps` counts global steps, like Lightning's."""
    n = start_step
    bs = getattr(loader, "batch_sampler", None)
    index_skip = isinstance(bs, SkippingBatchSampler)
    for ep in range(start_epoch, epochs):
        if index_skip:
            bs.set_epoch(ep)                 # a different shuffle per epoch: a function of (seed, epoch), on every rank
            bs.skip = skip if ep == start_epoch else 0
        elif sampler is not None:
            sampler.set_epoch(ep)
        for bi, batch in enumerate(loader):
            if not index_skip and ep == start_epoch and bi < skip:
                continue                     # (a loader of another kind: the consumed batches are loaded and dropped)
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
    task.model.engine.seed = int(cfg.seed)
    with_prev = "WithPrev" in type(task).__name__
    synthetic = bool(cfg.get("synthetic", False))
    max_steps = cfg.get("max_steps")
    max_steps = None if max_steps is None else int(max_steps)
    train_loader = val_loader = sampler = None
    if not synthetic:
        train_loader, val_loader, sampler = real_loaders(cfg, world, rank)
    start, start_epoch, skip, resumed = 0, 0, 0, False
    path = cfg.get("path")
    if path is not None and str(path) != "":
        path = str(path)
        if path.endswith(".ckpt"):
            start = trainer.resume(path)
            if train_loader is not None:
                start_epoch, skip = resume_position(start, len(train_loader))
            if rank == 0:
                print(f"Resuming from {path} at step {start}, epoch {start_epoch} (+{skip} batches)", flush=True)
        elif path.endswith(".pth"):
            if rank == 0:
                print(f"Loading weights from {path}...", flush=True)
            trainer.resume(path)
        else:
            raise ValueError(f"Invalid extension for path: {path}")
        resumed = True
    # every rank draws its own dropout masks (torch DDP: one generator per process), not world copies of one mask
    # (after the resume, which restores the seed rank 0 saved)
    task.model.engine.seed += 1000003 * rank
    log_every = max(1, int(cfg.trainer.get("log_every_n_steps", 100)))
    if synthetic:
        steps = 10 if max_steps is None else max_steps
        for it, (audio, labels, prev) in enumerate(synthetic_batches(cfg, rank, device, steps, with_prev), start):
            loss = trainer.train_step(audio, labels, prev, audio=True)
            if rank == 0 and (it % log_every == 0 or it == start + steps - 1):
                print(f"step {it} train_loss {loss.item():.4f}", flush=True)
    else:
        val_every = max(1, int(cfg.trainer.get("check_val_every_n_epoch", 1)))
        it, last_ep, loss = start, start_epoch, None

        def validate(ep):
            """val_loss = mean over the batches of ALL ranks (each rank holds a shard of the validation set)."""
            acc = torch.zeros(2, device=device, dtype=torch.float64)
            for b in val_loader:
                prev = b[2].to(device) if len(b) > 2 else None
                acc[0] += trainer.eval_loss(b[0].to(device), b[1].to(device), prev).double().sum()
                acc[1] += 1
            if world > 1:
                dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            if rank == 0 and acc[1].item() > 0:
                print(f"epoch {ep} val_loss {(acc[0] / acc[1]).item():.4f}", flush=True)

        if resumed:
            validate(start_epoch)        # the reference validates the loaded weights before fit (train.py:61-87)
        for ep, mel, labels, prev in loader_batches(train_loader, device, int(cfg.num_epochs), max_steps,
                                                    start_epoch, start, sampler, skip):
            if ep != last_ep and (last_ep + 1) % val_every == 0:
                validate(last_ep)
            last_ep = ep
            loss = trainer.train_step(mel, labels, prev, audio=False)
            if rank == 0 and it % log_every == 0:
                print(f"step {it} train_loss {loss.item():.4f}", flush=True)
            it += 1
        if rank == 0 and loss is not None and (it - 1) % log_every != 0:
            print(f"step {it - 1} train_loss {loss.item():.4f}", flush=True)
        if (last_ep + 1) % val_every == 0:
            validate(last_ep)
    if rank == 0:
        out_dir = os.path.join(str(cfg.get("output_dir", ".")), f"{cfg.model_type}_{cfg.dataset_type}",
                               "version_0", "checkpoints")
        os.makedirs(out_dir, exist_ok=True)
        trainer.save_checkpoint(os.path.join(out_dir, "last.ckpt"), epoch=0 if synthetic else last_ep)
        trainer.save_checkpoint(os.path.join(out_dir, "last.pt"))
        print(f"Saved model in {os.path.join(out_dir, 'last.pt')}.", flush=True)
    trainer.close()                      # captured graphs, then the library's own RCCL communicator (MRMT3_DDP_NATIVE), before torch's
    if world > 1:
        dist.destroy_process_group()
    return task


if __name__ == "__main__":
    main()
