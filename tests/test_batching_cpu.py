"""Host half of the GPU-resident batch construction (mrmt3.batching) against the literal restatement of
dataset/dataset_2_random.py:292-344,395-400 in oracle/logmel_ref.py, with the same `random` seeds."""
import random

import numpy as np
import pytest

from mrmt3.batching import pad_targets, plan_crops
from oracle import logmel_ref as ref


def _oracle_plan(n_frames, mel_length, rows_per_batch, split_len, seed, deterministic=False):
    rng = random.Random(seed)
    row = {"inputs": np.arange(n_frames)[:, None] * np.ones((1, 2)), "input_times": np.arange(n_frames),
           "targets": np.zeros(3)}
    rows = ref.select_rows(ref.split_frame(row, split_len), rows_per_batch, rng, deterministic)
    out = []
    for r in rows:
        c = ref.random_chunk(r, mel_length, rng, deterministic)
        out.append((int(c["inputs"][0, 0]), min(mel_length, c["inputs"].shape[0])))
    return out


@pytest.mark.parametrize("n_frames,split_len,rows", [(30000, 2000, 12), (4001, 2000, 12), (4000, 2000, 12),
                                                       (1999, 2000, 12), (100, 2000, 12), (256, 2000, 12),
                                                       (257, 2000, 12), (9000, 256, 4), (50000, 2000, 3)])
@pytest.mark.parametrize("deterministic", [False, True])
def test_crop_plan_matches_reference_arithmetic(n_frames, split_len, rows, deterministic):
    for seed in (0, 1, 365):
        want = _oracle_plan(n_frames, 256, rows, split_len, seed, deterministic)
        got = plan_crops(n_frames, 256, rows, split_len, deterministic, rng=random.Random(seed))
        assert list(zip(got.start_frame.tolist(), got.valid_frames.tolist())) == want
        assert (got.start_frame + got.valid_frames <= n_frames).all()


def test_crop_plan_known_answers():
    # 4000 frames, chunk 2000: range(0,4000,2000) -> 0 kept (0+2000 < 4000), 2000 dropped (>=) -> 1 row
    assert len(plan_crops(4000, rng=random.Random(0)).start_frame) == 1
    # 4001 frames: chunks at 0 and 2000 kept, 4000 dropped
    p = plan_crops(4001, rng=random.Random(0))
    assert p.chunk_start.tolist() == [0, 2000] and all(0 <= s - c <= 2000 - 256 for s, c in zip(p.start_frame, p.chunk_start))
    # shorter than a window: one row, everything valid, no draw
    p = plan_crops(100, rng=None)
    assert p.start_frame.tolist() == [0] and p.valid_frames.tolist() == [100]
    # deterministic: first run of chunks, offset 0
    p = plan_crops(50000, num_rows_per_batch=3, is_deterministic=True)
    assert p.start_frame.tolist() == [0, 2000, 4000]


def test_pad_targets_matches_reference():
    rng = np.random.RandomState(0)
    tg = [rng.randint(0, 1300, size=n) for n in (0, 1, 10, 1022, 1023, 1024, 1500)]
    got = pad_targets(tg, 1024).numpy()
    for i, t in enumerate(tg):
        _, want = ref.pad_length(np.zeros((256, 4), np.float32), t, 256, 1024)
        assert want.shape == (1024,)
        np.testing.assert_array_equal(got[i], want)
    assert got[0, 0] == 1 and got[0, 1] == -100                 # empty target: EOS then padding
    assert got[4, 1023] == 1 and got[5, 1023] == tg[5][1023] + 3   # 1023 tokens: EOS fits; 1024: no EOS


def test_pad_length_mel_half():
    mel = np.ones((57, 512), np.float32)
    m, _ = ref.pad_length(mel, np.zeros(2), 256, 1024)
    assert m.shape == (256, 512) and m[:57].all() and not m[57:].any()
