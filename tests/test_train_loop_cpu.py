"""Host logic of mr-mt3_amd/train.py that needs no GPU: the data loaders under several ranks (what Lightning's DDP
strategy does to the reference's plain DataLoaders, config/config.yaml:45) and the resume arithmetic of the batch
iterator."""
import sys

import torch


def _cfg(tmp_path, n_train=10, n_val=4):
    from mrmt3 import hydra_lite
    (tmp_path / "toyset2.py").write_text(
        "import torch\n"
        "from torch.utils.data import Dataset\n"
        "class Toy(Dataset):\n"
        "    def __init__(self, n): self.n = n\n"
        "    def __len__(self): return self.n\n"
        "    def __getitem__(self, i): return torch.full((1, 4), float(i)), torch.full((1, 2), i, dtype=torch.int64)\n"
        "def collate(batch):\n"
        "    return torch.cat([b[0] for b in batch]), torch.cat([b[1] for b in batch])\n")
    if str(tmp_path) not in sys.path:
        sys.path.insert(0, str(tmp_path))
    return hydra_lite._wrap({
        "seed": 365,
        "dataset": {"train": {"_target_": "toyset2.Toy", "n": n_train}, "val": {"_target_": "toyset2.Toy", "n": n_val},
                    "collate_fn": "toyset2.collate"},
        "dataloader": {"train": {"batch_size": 2, "shuffle": True}, "val": {"batch_size": 2, "shuffle": False}}})


def test_real_loaders_shard_the_datasets_over_the_ranks(tmp_path):
    import train
    cfg = _cfg(tmp_path)
    tl, vl, sampler = train.real_loaders(cfg)                     # one rank: plain loaders, no sampler
    assert sampler is None and sum(b[0].shape[0] for b in tl) == 10
    seen, val_seen = [], []
    for rank in range(2):
        tl, vl, sampler = train.real_loaders(cfg, world=2, rank=rank)
        assert sampler is not None
        sampler.set_epoch(0)
        seen.append(sorted(int(v) for b in tl for v in b[1][:, 0]))
        val_seen.append(sorted(int(v) for b in vl for v in b[1][:, 0]))
    # every rank gets half of the samples, together they cover the set exactly once (10 is divisible by 2: no padding)
    assert len(seen[0]) == len(seen[1]) == 5 and sorted(seen[0] + seen[1]) == list(range(10))
    assert sorted(val_seen[0] + val_seen[1]) == list(range(4))
    # another epoch, another shuffle — the same one on both ranks (still a partition)
    orders = []
    for rank in range(2):
        tl, _, sampler = train.real_loaders(cfg, world=2, rank=rank)
        sampler.set_epoch(1)
        orders.append([int(v) for b in tl for v in b[1][:, 0]])
    assert sorted(orders[0] + orders[1]) == list(range(10))


def test_loader_batches_counts_global_steps_and_resumes_mid_run(tmp_path):
    import train
    cfg = _cfg(tmp_path, n_train=6)
    tl, _, _ = train.real_loaders(cfg)                            # 3 batches per epoch
    dev = torch.device("cpu")
    full = [(ep, int(x[0, 0])) for ep, x, _, _ in train.loader_batches(tl, dev, epochs=2, max_steps=None)]
    assert [e for e, _ in full] == [0, 0, 0, 1, 1, 1]
    capped = list(train.loader_batches(tl, dev, epochs=2, max_steps=4))
    assert len(capped) == 4
    # resumed at global step 3 in epoch 1 with max_steps = 5: two more steps, all in epoch 1
    resumed = list(train.loader_batches(tl, dev, epochs=2, max_steps=5, start_epoch=1, start_step=3))
    assert [b[0] for b in resumed] == [1, 1]


def test_resume_position_from_the_step_count_never_overshoots_the_schedule(tmp_path):
    """a run resumed from its own last.ckpt must continue where it stopped — epoch and batch offset derived from
    the global step — and take exactly the remaining steps of num_epochs x steps_per_epoch, whether the checkpoint was
    written mid-epoch or at an epoch's end."""
    import train
    cfg = _cfg(tmp_path, n_train=6)
    tl, _, _ = train.real_loaders(cfg)                            # 3 batches per epoch
    dev = torch.device("cpu")
    spe, epochs = len(tl), 3
    assert spe == 3
    assert train.resume_position(0, spe) == (0, 0)
    assert train.resume_position(4, spe) == (1, 1)                # mid-epoch: one batch of epoch 1 already consumed
    assert train.resume_position(6, spe) == (2, 0)                # end of epoch 1: continue with epoch 2, nothing skipped
    assert train.resume_position(9, spe) == (3, 0)                # the schedule is finished: no epoch left
    for done in range(0, spe * epochs + 1):
        ep0, skip = train.resume_position(done, spe)
        rest = list(train.loader_batches(tl, dev, epochs=epochs, max_steps=None, start_epoch=ep0, start_step=done, skip=skip))
        assert done + len(rest) == spe * epochs, (done, len(rest))
        assert [b[0] for b in rest] == [e for e in range(epochs) for _ in range(spe)][done:]


def test_resumed_epoch_replays_the_interrupted_order_without_loading_the_consumed_batches(tmp_path):
    """at ONE rank too the epoch's order is a function of (seed, epoch) — the resumed epoch continues the
    interrupted one sample for sample — and the consumed batches are skipped at the index level: the dataset is never
    asked for them."""
    import train
    cfg = _cfg(tmp_path, n_train=12)
    dev = torch.device("cpu")
    tl, _, _ = train.real_loaders(cfg)
    spe = len(tl)
    assert spe == 6
    torch.manual_seed(1)
    full = [[int(v) for v in y[:, 0]] for _, _, y, _ in train.loader_batches(tl, dev, epochs=3, max_steps=None)]
    assert sorted(v for b in full[:spe] for v in b) == list(range(12))                # an epoch is a permutation ...
    assert full[:spe] != full[spe:2 * spe]                                              # ... another one every epoch
    torch.manual_seed(999)                                                              # (the global RNG plays no part)
    tl2, _, _ = train.real_loaders(cfg)
    fetched = []
    ds = tl2.dataset
    real_get = type(ds).__getitem__
    type(ds).__getitem__ = lambda self, i: (fetched.append(i), real_get(self, i))[1]
    try:
        for done in (0, 4, 6, 8, 13):
            del fetched[:]
            ep0, skip = train.resume_position(done, spe)
            rest = [[int(v) for v in y[:, 0]] for _, _, y, _ in
                    train.loader_batches(tl2, dev, epochs=3, max_steps=None, start_epoch=ep0, start_step=done, skip=skip)]
            assert rest == full[done:], done
            assert sorted(fetched) == sorted(v for b in full[done:] for v in b), done   # nothing consumed earlier was loaded
    finally:
        type(ds).__getitem__ = real_get
