"""The element-wise dropout mask generator (oracle/dropout_ref.py = csrc/common.h): distribution checks on the CPU.
The GPU kernels are compared with the same restatement bit for bit in test_kernels_gpu.py."""
import numpy as np
import pytest

from oracle import dropout_ref as dr


@pytest.mark.parametrize("seed,stream", [(1234, 5), (0xDEADBEEFCAFE, 17), (7, 0), (2 ** 63 + 12345, 3)])
def test_keep_rate_histogram_and_correlations(seed, stream):
    rows, cols, p = 2048, 512, 0.1
    keep, scale = dr.keep_mask(rows * cols, p, seed, stream)
    assert abs(scale - 65536.0 / (65536.0 - 6554.0)) < 1e-6
    k = keep.reshape(rows, cols).astype(np.float64)
    rate = k.mean()
    assert abs(rate - (1.0 - 6554.0 / 65536.0)) < 1.5e-3                      # 1M samples: sigma = 3e-4
    assert abs(rate * scale - 1.0) < 2e-3                                      # unbiased in expectation
    kf = k - rate
    for lag in (1, 2, 3, 4, 8):
        assert abs((kf[:, lag:] * kf[:, :-lag]).mean() / kf.var()) < 5e-3     # along a row
    for lag in (1, 2, 4):
        assert abs((kf[lag:] * kf[:-lag]).mean() / kf.var()) < 5e-3           # across rows
    # keep counts per row / per column are binomial
    assert 0.8 < k.sum(1).var() / (cols * rate * (1 - rate)) < 1.25
    assert 0.8 < k.sum(0).var() / (rows * rate * (1 - rate)) < 1.25


def test_sites_and_seeds_are_independent_and_p_zero_keeps_everything():
    n = 1 << 20
    a, _ = dr.keep_mask(n, 0.1, 1234, 5)
    b, _ = dr.keep_mask(n, 0.1, 1234, 6)          # another dropout site of the same step
    c, _ = dr.keep_mask(n, 0.1, 1235, 5)          # the next step's seed
    af = a - a.mean()
    for other in (b, c):
        assert abs((af * (other - other.mean())).mean() / af.var()) < 5e-3
        assert (a != other).mean() > 0.15          # 2 p (1 - p) = 0.18 for independent masks
    keep, scale = dr.keep_mask(4096, 0.0, 1, 1)
    assert keep.all() and scale == 1.0
    for p in (0.05, 0.3, 0.5):
        k, s = dr.keep_mask(n, p, 99, 2)
        assert abs(k.mean() - (1 - p)) < 3e-3 and abs(k.mean() * s - 1.0) < 4e-3


def test_step_salt_gives_independent_masks_per_step():
    """The in-kernel step salt (device step counter, hipGraph replays): consecutive steps draw independent masks with
    the same keep rate; no counter attached = the unsalted mask."""
    n = 1 << 20
    base, _ = dr.keep_mask(n, 0.1, 365, 9)
    same, _ = dr.keep_mask(n, 0.1, 365, 9, step=None)
    assert (base == same).all()
    masks = [dr.keep_mask(n, 0.1, 365, 9, step=s)[0] for s in (0, 1, 2, 3, 64500)]
    for i, a in enumerate(masks):
        assert abs(a.mean() - 0.9) < 2e-3
        assert (a != base).mean() > 0.15
        af = a - a.mean()
        for b in masks[i + 1:]:
            assert abs((af * (b - b.mean())).mean() / af.var()) < 5e-3
    ka, _ = dr.attn_keep_mask(1, 2, 128, 128, 0.1, 365, 4, step=5)
    kb, _ = dr.attn_keep_mask(1, 2, 128, 128, 0.1, 365, 4, step=6)
    assert abs(ka.mean() - (1 - 26 / 256)) < 1e-2 and (ka != kb).mean() > 0.12


@pytest.mark.parametrize("seed,stream", [(365, 1), (1234, 7), (0xDEADBEEFCAFE, 17)])
def test_attention_mask_keep_rate_and_correlations(seed, stream):
    """The attention-probability mask (one mix per query row and four consecutive keys, a byte per element): keep rate
    1 - 26/256, no correlation along a query row (lags inside and across the four-key groups), down a key column,
    between heads or between batch rows; keep counts per row and per column binomial."""
    keep, scale = dr.attn_keep_mask(2, 6, 512, 512, 0.1, seed, stream)
    assert abs(scale - 256.0 / 230.0) < 1e-9
    k = keep.astype(np.float64)
    rate = k.mean()
    assert abs(rate - (1.0 - 26.0 / 256.0)) < 1.5e-3                          # 3.1M samples: sigma = 1.7e-4
    kf = k - rate
    var = kf.var()
    for lag in (1, 2, 3, 4, 5, 8, 16):
        assert abs((kf[..., :, lag:] * kf[..., :, :-lag]).mean() / var) < 3e-3   # along a query row
    for lag in (1, 2, 3, 4, 8, 16):
        assert abs((kf[..., lag:, :] * kf[..., :-lag, :]).mean() / var) < 3e-3   # down a key column
    assert abs((kf[:, 1:] * kf[:, :-1]).mean() / var) < 3e-3                     # neighbouring heads
    assert abs((kf[1:] * kf[:-1]).mean() / var) < 3e-3                           # neighbouring batch rows
    assert 0.9 < k.sum(-1).var() / (512 * rate * (1 - rate)) < 1.1
    assert 0.9 < k.sum(-2).var() / (512 * rate * (1 - rate)) < 1.1
    other, _ = dr.attn_keep_mask(2, 6, 512, 512, 0.1, seed, stream + 1)         # another attention site of the step
    assert abs((kf * (other - other.mean())).mean() / var) < 3e-3 and (keep != other).mean() > 0.15
