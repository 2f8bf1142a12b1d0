"""Kernel-level parity on a real MI355X: every C-ABI entry point against a plain fp32 PyTorch
restatement of the same op (or the oracle) on identical seeded inputs.  Tolerances are stated per
test; integer / index work is bit-exact."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mrmt3 import lib
    lib.load()
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


# ---- K1 log-mel -------------------------------------------------------------------------------------

def test_logmel_matches_oracle(dev):
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio
    from oracle import logmel_ref
    audio = synth_audio(3)
    ref = logmel_ref.logmel_segments(audio)                      # [3,256,512] normalised
    got = sp.logmel_segments(torch.from_numpy(audio).to(dev)).cpu().numpy()
    assert got.shape == (3, 256, 512)
    # tolerance: 1e-4 absolute on the [0,1]-normalised log-mel
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)
    # un-normalised API + the reference's numpy signature
    one = sp.compute_spectrogram(audio[0], sp.SpectrogramConfig())
    np.testing.assert_allclose(one, logmel_ref.compute_spectrogram(audio[0]), atol=2e-3, rtol=0)


def test_logmel_edge_cases(dev):
    from contrib import spectrograms as sp
    from oracle import logmel_ref
    rs = np.random.RandomState(1)
    # ragged length (not a hop multiple), silence, and a pure tone
    x = rs.uniform(-1, 1, size=(1, 40000)).astype(np.float32)
    got = sp.logmel_segments(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = logmel_ref.logmel_segments(x)
    assert got.shape == ref.shape == (1, 313, 512)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)
    z = np.zeros((1, 32768), np.float32)
    got = sp.logmel_segments(torch.from_numpy(z).to(dev)).cpu().numpy()
    assert np.allclose(got, (math.log(1e-5) + 12) / 17, atol=1e-6)      # safe_log floor everywhere
    t = np.arange(32768) / 16000.0
    tone = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)[None]
    got = sp.logmel_segments(torch.from_numpy(tone).to(dev), normalize=False).cpu().numpy()
    ref = np.stack([logmel_ref.compute_spectrogram(tone[0])])
    assert got[0, 10].argmax() == ref[0, 10].argmax()
    big = ref > ref.max() - 8.0                                         # fp32 FFT noise floor below that
    np.testing.assert_allclose(got[big], ref[big], atol=2e-3, rtol=0)
    # valid_frames zeroing (inference.py:125-126) and bf16 output
    vf = torch.tensor([100], dtype=torch.int32, device=dev)
    g = sp.logmel_segments(torch.from_numpy(x[:, :32768]).to(dev), valid_frames=vf).cpu().numpy()
    assert (g[0, 100:] == 0).all() and (g[0, :100] != 0).any()
    gb = sp.logmel_segments(torch.from_numpy(x[:, :32768]).to(dev), out_bf16=True).float().cpu().numpy()
    np.testing.assert_allclose(gb, logmel_ref.logmel_segments(x[:, :32768]), atol=4e-3, rtol=0)


def test_logmel_ignores_filter_taps_beyond_fb_cnt_and_takes_unaligned_tables(dev, knobs):
    """include/mrmt3_hip.h: filter m = sum over q < fb_cnt[m]; whatever fb_w holds behind a filter's last tap (NaN,
    garbage: a C-ABI caller need not zero-pad) is never used — by either kernel.  Tables that are not
    aligned for the wave-per-frame kernel's vector reads go to the general kernel, same numbers."""
    from contrib import spectrograms as sp
    from mrmt3 import lib
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(-1, 1, size=(3, 32768)).astype(np.float32)).to(dev)
    t = dict(sp.kernel_tables(sp.SpectrogramConfig(), dev))
    want = lib.logmel(x, t)
    bad = dict(t)
    w = t["fb_w"].clone()
    q = torch.arange(t["max_taps"], device=dev)[None, :]
    tail = q >= t["fb_cnt"][:, None]
    assert tail.any()
    w[tail] = float("nan")
    w[tail & (q % 2 == 0)] = 1e30
    bad["fb_w"] = w
    for mode in (1, 0):
        knobs.set("MRMT3_LOGMEL", mode)
        got = lib.logmel(x, bad)
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() <= (0.0 if mode == 1 else 2e-4), mode
    knobs.set("MRMT3_LOGMEL", 1)
    # fb_start 4 bytes off a 16-byte boundary, the window 4 bytes off an 8-byte one: the general kernel takes the call
    off = dict(t)
    buf = torch.zeros(t["fb_start"].numel() + 1, dtype=torch.int32, device=dev)
    buf[1:] = t["fb_start"]
    off["fb_start"] = buf[1:]
    wbuf = torch.zeros(t["window"].numel() + 1, device=dev)
    wbuf[1:] = t["window"]
    off["window"] = wbuf[1:]
    assert off["fb_start"].data_ptr() % 16 == 4 and off["window"].data_ptr() % 8 == 4
    got = lib.logmel(x, off)
    assert (got - want).abs().max().item() <= 2e-4


def test_logmel_wave_kernel_matches_the_round1_kernel(dev, knobs):
    """Round 4's wave-per-frame kernel (64 frames per workgroup, a 16 x 16 x 4 FFT in registers, csrc/logmel.hip) against
    round 1's workgroup-per-frame kernel (MRMT3_LOGMEL=0): two independent implementations of the same transform
    (contrib/spectrograms.py:128-145) agree to f32 FFT rounding on ragged lengths, frame counts off the 64-frame grid,
    padded frames, bf16 output and crops at odd sample offsets; the mel sums and the scaling are the same arithmetic."""
    from contrib import spectrograms as sp
    rs = np.random.RandomState(11)
    for B, n in ((2, 32768), (1, 40000), (3, 128 * 70 + 5), (1, 2048), (1, 300), (5, 128 * 129)):
        x = torch.from_numpy((rs.uniform(-1, 1, size=(B, n)) * rs.uniform(0.01, 1.0, size=(B, 1))).astype(np.float32)).to(dev)
        vf = torch.tensor([max(1, (n // 128) // (b + 2)) for b in range(B)], dtype=torch.int32, device=dev)
        for kw in (dict(), dict(valid_frames=vf), dict(normalize=False), dict(out_bf16=True)):
            knobs.set("MRMT3_LOGMEL", "0")
            old = sp.logmel_segments(x, **kw).float()
            knobs.set("MRMT3_LOGMEL", "1")
            new = sp.logmel_segments(x, **kw).float()
            assert new.shape == old.shape
            tol = 4e-3 if kw.get("out_bf16") else (2e-4 if kw.get("normalize", True) else 3e-3)
            # un-normalised: log of near-cancelled bins amplifies the FFT's rounding; bound where the energy is
            d = (new - old).abs()
            if kw.get("normalize", True):
                assert d.max().item() <= tol, (B, n, kw, d.max().item())
            else:
                big = old > old.max() - 10.0
                assert d[big].max().item() <= tol, (B, n, kw, d[big].max().item())
            if "valid_frames" in kw:
                for b in range(B):
                    assert (new[b, int(vf[b]):] == 0).all()
    # crops out of one recording at odd sample offsets
    song = torch.from_numpy(rs.uniform(-1, 1, size=128 * 700 + 77).astype(np.float32)).to(dev)
    starts = torch.tensor([0, 13, 255, 600], dtype=torch.int64)
    vfc = torch.tensor([256, 200, 256, 64], dtype=torch.int32)
    knobs.set("MRMT3_LOGMEL", "0")
    old = sp.logmel_crops(song, starts, 256, valid_frames=vfc)
    knobs.set("MRMT3_LOGMEL", "1")
    new = sp.logmel_crops(song, starts, 256, valid_frames=vfc)
    assert (new - old).abs().max().item() <= 2e-4
    assert (new[1, 200:] == 0).all() and (new[3, 64:] == 0).all()


# ---- GEMMs ------------------------------------------------------------------------------------------

def test_logmel_crops_of_one_recording(dev):
    """GPU-resident batch construction (mrmt3.batching): crops read straight out of the recording ==
    the oracle's per-crop `_compute_spectrogram` + `_pad_length`, and == materialised crops bit for bit."""
    import random
    from contrib import spectrograms as sp
    from mrmt3.batching import CropPlan, DeviceBatcher
    from oracle import logmel_ref
    rs = np.random.RandomState(7)
    song = rs.uniform(-1, 1, size=9000 * 128 + 37).astype(np.float32)          # not a whole number of frames
    bt = DeviceBatcher(dev, mel_length=256, num_rows_per_batch=3, split_frame_length=2000, rng=random.Random(5))
    audio = bt.upload(song)
    assert audio.numel() == 9001 * 128
    plan = bt.plan(audio.numel() // 128)
    assert len(plan.start_frame) == 3 and (plan.valid_frames == 256).all()
    mel, tg = bt.build(audio, lambda s, n: np.arange(5) + s % 7, plan)
    assert mel.shape == (3, 256, 512) and tg.shape == (3, 1024) and tg.device.type == "cuda"
    assert tg[0, :7].tolist() == [3 + plan.start_frame[0] % 7 + i for i in range(5)] + [1, -100]
    padded = np.concatenate([song, np.zeros(128 - 37, np.float32)])
    for b, s in enumerate(plan.start_frame):
        frames = padded[s * 128:(s + 256) * 128].reshape(256, 128)
        want, _ = logmel_ref.pad_length(logmel_ref.compute_spectrogram_row(frames), np.zeros(1), 256, 1024)
        np.testing.assert_allclose(mel[b].cpu().numpy(), want, atol=1e-4, rtol=0)
    crops = torch.stack([audio[s * 128:(s + 256) * 128] for s in plan.start_frame])
    assert torch.equal(mel, sp.logmel_segments(crops))
    # a crop never sees its neighbour: zeroing everything outside it changes nothing
    s0 = int(plan.start_frame[0])
    alone = torch.zeros_like(audio)
    alone[s0 * 128:(s0 + 256) * 128] = audio[s0 * 128:(s0 + 256) * 128]
    one = CropPlan(plan.start_frame[:1], plan.valid_frames[:1], plan.chunk_start[:1])
    assert torch.equal(bt.mel(alone, one)[0], mel[0])
    # recording shorter than one window: its 100 frames are transformed on their own, the rest is zero rows
    short = bt.upload(song[:100 * 128])
    p = bt.plan(100)
    m = bt.mel(short, p)
    want, _ = logmel_ref.pad_length(logmel_ref.compute_spectrogram_row(song[:100 * 128].reshape(100, 128)),
                                    np.zeros(1), 256, 1024)
    np.testing.assert_allclose(m[0].cpu().numpy(), want, atol=1e-4, rtol=0)
    assert (m[0, 100:] == 0).all()
    # a crop running past the end of the recording reads zeros there, not out of bounds
    tail = CropPlan(np.asarray([9001 - 40], np.int64), np.asarray([256], np.int32), np.asarray([0], np.int64))
    mt = bt.mel(audio, tail)
    ref_tail = np.zeros(256 * 128, np.float32)
    ref_tail[:40 * 128] = padded[(9001 - 40) * 128:]
    np.testing.assert_allclose(mt[0].cpu().numpy(), logmel_ref.compute_spectrogram_row(ref_tail.reshape(256, 128)),
                               atol=1e-4, rtol=0)
    # bf16 output feeds the trainer directly
    btb = DeviceBatcher(dev, num_rows_per_batch=3, out_bf16=True)
    assert btb.mel(audio, plan).dtype == torch.bfloat16


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 1152, 512), (4096, 512, 384), (130, 2048, 512),
                                   (2048, 512, 1024), (64, 1536, 512)])
def test_gemm_nt_bf16(dev, M, N, K):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = torch.randn(N, K, generator=g).to(dev).bfloat16()
    ref = a.float() @ b.float().t()
    out32 = lib.gemm_nt(a, b, out_dtype=torch.float32)
    assert _rel(out32, ref) < 1e-5                       # fp32 accumulation of exact bf16 products
    out16 = lib.gemm_nt(a, b, out_dtype=torch.bfloat16)
    assert _rel(out16, ref) < 4e-3                       # one bf16 rounding of the result
    acc = torch.ones(M, N, device=dev)
    lib.gemm_nt(a, b, out=acc, accumulate=True)
    assert _rel(acc, ref + 1.0) < 1e-5


def test_gemm_full_benchmark_sizes(dev):
    """The batch-64 shapes of the benchmark step (65 536 token rows): NT against an f32 matmul on sampled rows, the
    wgrad TN product in full, plus linearity (C(2A) == 2 C(A) exactly: powers of two commute with bf16 rounding)."""
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(1)
    M = 65536
    for N, K in ((2048, 512), (1152, 512), (512, 1024)):
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
        b = (torch.randn(N, K, generator=g) * 0.05).to(dev).bfloat16()
        out = lib.gemm_nt(a, b)
        rows = torch.randint(0, M, (512,), generator=g).to(dev)
        ref = a[rows].float() @ b.float().t()
        assert _rel(out[rows], ref) < 6e-3
        assert torch.equal(lib.gemm_nt(a * 2, b), out * 2)
    dy = (torch.randn(M, 1152, generator=g) * 0.1).to(dev).bfloat16()
    x = torch.randn(M, 512, generator=g).to(dev).bfloat16()
    dw = torch.zeros(1152, 512, device=dev)
    lib.gemm_tn(dy, x, dw, accumulate=True)
    ref = dy.float().t() @ x.float()
    assert _rel(dw, ref) < 2e-3
    dw2 = torch.zeros(1152, 512, device=dev)
    lib.gemm_tn(dy, x, dw2, accumulate=True)
    assert torch.equal(dw, dw2)                                    # fixed-order split-K: bitwise reproducible


def test_gemm_nt_strided_and_asymmetric(dev):
    from mrmt3 import lib
    # A = I (padded) with an asymmetric B catches a transposed C write
    K = 128
    a = torch.zeros(128, K, device=dev)
    a[:, :128] = torch.eye(128, device=dev)
    b = torch.arange(256 * K, device=dev, dtype=torch.float32).reshape(256, K) % 251
    out = lib.gemm_nt(a.bfloat16(), b.bfloat16(), out_dtype=torch.float32)
    assert torch.equal(out, b[:, :128].t().contiguous())
    # row-strided views (q slice of a fused qkv buffer)
    big = torch.randn(300, 1152, device=dev).bfloat16()
    w = torch.randn(512, 384, device=dev).bfloat16()
    out = lib.gemm_nt(big[:, 384:768], w, out_dtype=torch.float32)
    assert _rel(out, big[:, 384:768].float() @ w.float().t()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (777, 1152, 512), (300, 512, 1024)])
def test_gemm_nt_f32(dev, M, N, K):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.randn(M, K, generator=g).to(dev)
    b = torch.randn(N, K, generator=g).to(dev)
    ref = (a.double() @ b.double().t())
    out = lib.gemm_nt(a, b)
    assert _rel(out.double(), ref) < 2e-6                # exact-f32 MFMA chain


@pytest.mark.parametrize("M,N1,N2", [(512, 128, 128), (4096, 1152, 512), (1000, 512, 384), (8192, 2048, 512),
                                     (640, 1536, 512)])
def test_gemm_tn(dev, M, N1, N2):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(M)
    a = torch.randn(M, N1, generator=g).to(dev).bfloat16()
    b = torch.randn(M, N2, generator=g).to(dev).bfloat16()
    ref = a.float().t() @ b.float()
    out = torch.full((N1, N2), 7.0, device=dev)
    lib.gemm_tn(a, b, out)
    assert _rel(out, ref) < 1e-5
    out2 = torch.ones(N1, N2, device=dev)
    lib.gemm_tn(a, b, out2, accumulate=True)
    assert _rel(out2, ref + 1.0) < 1e-5
    out3 = torch.empty(N1, N2, device=dev)
    lib.gemm_tn(a, b, out3)
    assert torch.equal(out, out3)                        # fixed-order slab reduction: bitwise reproducible


# ---- RMS norm ----------------------------------------------------------------------------------------

def _rms_ref(x, w, eps=1e-6):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
def test_add_rmsnorm_fwd_bwd(dev, ydt):
    from mrmt3 import lib
    rows, cols = 1000, 512
    x0 = torch.randn(rows, cols, device=dev)
    y = torch.randn(rows, cols, device=dev).to(ydt)
    w = (1 + 0.1 * torch.randn(cols, device=dev))
    x1, xn, rstd = lib.add_rmsnorm_fwd(x0, y, w, 1e-6, torch.float32)
    x1r = (x0 + y.float()).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    xnr = _rms_ref(x1r, wr)
    assert torch.allclose(x1, x1r, atol=1e-6) and torch.allclose(xn, xnr, atol=2e-6, rtol=1e-5)
    _, xnb, _ = lib.add_rmsnorm_fwd(x0, y, w, 1e-6, torch.bfloat16)
    assert _rel(xnb, xnr) < 4e-3
    _, xn0, _ = lib.add_rmsnorm_fwd(x0, None, w, 1e-6, torch.float32, write_x1=False)
    assert torch.allclose(xn0, _rms_ref(x0, w), atol=2e-6, rtol=1e-5)
    # backward
    dxn = torch.randn(rows, cols, device=dev)
    dres = torch.randn(rows, cols, device=dev)
    (xnr * dxn).sum().backward()
    dw = torch.zeros(cols, device=dev)
    dx1, dy = lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, w, dw)
    assert torch.allclose(dx1, x1r.grad + dres, atol=1e-5, rtol=1e-4)
    assert _rel(dy, x1r.grad + dres) < 4e-3
    assert _rel(dw, wr.grad) < 1e-5


def test_norm_weight_gradients_batched_over_sites(dev):
    """`NormDwBatch`: several norm sites leave their partial rows in workspaces of their own and ONE launch sums them
    (mrmt3_norm_dw_reduce).  Same summation order as the immediate form -> bit-identical, ragged row counts, a second
    pass accumulates onto the first, and a site that is queued twice before a flush would be a bug we guard against
    by giving every (dw, shape) its own workspace."""
    from mrmt3 import lib
    cols = 512
    sites = []
    for i, rows in enumerate((1000, 64 * 256, 37, 4096)):
        g = torch.Generator(device="cpu").manual_seed(i)
        x1 = torch.randn(rows, cols, generator=g).to(dev)
        rstd = (torch.rand(rows, generator=g) + 0.5).to(dev)
        w = (1 + 0.1 * torch.randn(cols, generator=g)).to(dev)
        dxn = torch.randn(rows, cols, generator=g).to(dev)
        sites.append((dxn, x1, rstd, w))
    want = []
    for dxn, x1, rstd, w in sites:
        dw = torch.zeros(cols, device=dev)
        lib.add_rmsnorm_bwd(dxn, None, x1, rstd, w, dw, want_dy=False)
        want.append(dw)
    batch = lib.NormDwBatch()
    got = [torch.zeros(cols, device=dev) for _ in sites]
    for (dxn, x1, rstd, w), dw in zip(sites, got):
        lib.add_rmsnorm_bwd(dxn, None, x1, rstd, w, dw, want_dy=False, defer=batch)
    assert all(float(d.abs().max()) == 0.0 for d in got)        # nothing is summed before the flush
    batch.flush()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    for (dxn, x1, rstd, w), dw in zip(sites, got):               # second backward pass: accumulates, tables are reused
        lib.add_rmsnorm_bwd(dxn, None, x1, rstd, w, dw, want_dy=False, defer=batch)
    batch.flush()
    batch.flush()                                                # empty flush is a no-op
    for (dxn, x1, rstd, w), a, b in zip(sites, got, want):
        lib.add_rmsnorm_bwd(dxn, None, x1, rstd, w, b, want_dy=False)
        assert torch.equal(a, b)


@pytest.mark.parametrize("p,seed,stream", [(0.1, 1234, 5), (0.3, 2 ** 40 + 17, 1), (0.1, 2 ** 63 + 5, 200)])
def test_dropout_mask_equals_the_restatement_bit_for_bit(dev, p, seed, stream):
    """csrc/common.h drop_mask4 == oracle/dropout_ref.keep_mask: same kept elements, same keep scale."""
    from mrmt3 import lib
    from oracle import dropout_ref as dr
    rows, cols = 777, 512
    out = lib.dropmask_cast(torch.ones(rows, cols, device=dev), p=p, seed=seed, stream_id=stream).float().cpu().numpy().reshape(-1)
    keep, scale = dr.keep_mask(rows * cols, p, seed, stream)
    assert ((out != 0) == keep).all()
    assert abs(float(out[keep][0]) - scale) < 8e-3 * scale        # bf16 output
    # with a DEVICE step counter attached the kernel salts the key itself (hipGraph replays: by-value arguments are
    # frozen, the counter is not): every step a new mask, each equal to the restatement
    step = torch.zeros(1, device=dev, dtype=torch.int32)
    seen = [keep]
    for n in (0, 1, 2, 1000003):
        step.fill_(n)
        out = lib.dropmask_cast(torch.ones(rows, cols, device=dev), p=p, seed=seed, stream_id=stream,
                                step=step).float().cpu().numpy().reshape(-1)
        keep_n, _ = dr.keep_mask(rows * cols, p, seed, stream, step=n)
        assert ((out != 0) == keep_n).all()
        assert all((keep_n != k).mean() > 0.1 for k in seen)     # 2 p (1 - p) >= 0.18 for independent masks
        seen.append(keep_n)


def test_dropout_sites_consistent(dev):
    """Masks are a pure function of (seed, stream, index): forward and backward agree, keep rate is
    1-p, kept values are scaled by 1/(1-p)."""
    from mrmt3 import lib
    rows, cols, p = 2048, 512, 0.1
    x0 = torch.zeros(rows, cols, device=dev)
    y = torch.ones(rows, cols, device=dev)
    w = torch.ones(cols, device=dev)
    x1, _, rstd = lib.add_rmsnorm_fwd(x0, y, w, 1e-6, torch.float32, p=p, seed=1234, stream_y=5)
    keep = (x1 != 0)
    assert abs(keep.float().mean().item() - 0.9) < 2e-3
    assert torch.allclose(x1[keep], torch.full_like(x1[keep], 1 / 0.9))
    m2 = lib.dropmask_cast(torch.ones(rows, cols, device=dev), p=p, seed=1234, stream_id=5).float()
    assert torch.equal(m2 != 0, keep)
    m3 = lib.dropmask_cast(torch.ones(rows, cols, device=dev), p=p, seed=1234, stream_id=6).float()
    assert not torch.equal(m3 != 0, keep)


# ---- attention ---------------------------------------------------------------------------------------

def _attn_ref(q, k, v, B, H, Lq, Lk, causal):
    qh = q.float().view(B, Lq, H, 64).transpose(1, 2)
    kh = k.float().view(B, Lk, H, 64).transpose(1, 2)
    vh = v.float().view(B, Lk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(2, 3)
    if causal:
        i = torch.arange(Lq, device=q.device)[:, None]
        j = torch.arange(Lk, device=q.device)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B * Lq, H * 64), torch.logsumexp(s, -1)


ATTN_SHAPES = [(2, 6, 256, 256, False), (2, 6, 1024, 1024, True), (1, 6, 1024, 320, False),
               (2, 3, 200, 72, False), (1, 2, 300, 300, True), (1, 6, 64, 1024, False), (1, 1, 1088, 1088, True),
               (1, 1, 1, 1, False), (1, 2, 129, 257, False), (1, 1, 33, 33, True), (1, 2, 2048, 2112, False),
               # enough (batch, head) pairs for the PAIRED causal instantiations (>= 512 tile pairs; smaller causal
               # launches run their tiles unpaired), with an odd number of 128-row tiles
               (22, 6, 1024, 1024, True), (36, 6, 640, 640, True)]


@pytest.fixture(params=["coarse", "fine"])
def tile_rows(request, knobs):
    """128-row tiles (32 rows per wave) or the 64-row tiles small launches take (attn_fine, csrc/attn_common.h): every
    attention shape below runs both instantiations (paired causal launches are coarse either way)."""
    knobs.set("MRMT3_ATTN_FINE", "1" if request.param == "fine" else "0")
    return request.param


@pytest.mark.parametrize("B,H,Lq,Lk,causal", ATTN_SHAPES)
def test_attn_fwd_bf16(dev, tile_rows, B, H, Lq, Lk, causal):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(Lq * 7 + Lk)
    q = (torch.randn(B * Lq, H * 64, generator=g) * 0.35).to(dev).bfloat16()
    k = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    v = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    o, lse = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal)
    oref, lref = _attn_ref(q, k, v, B, H, Lq, Lk, causal)
    assert torch.allclose(lse, lref, atol=2e-3, rtol=1e-4)
    assert _rel(o, oref) < 1e-2 and (o.float() - oref).abs().max() < 3e-2


def test_attn_fwd_fused_qkv_layout(dev):
    """q/k/v as column slices of one [rows, 1152] buffer (the fused-QKV GEMM output)."""
    from mrmt3 import lib
    B, H, L = 2, 6, 256
    qkv = torch.randn(B * L, 1152, device=dev).bfloat16()
    qkv[:, :384] *= 0.35
    o, _ = lib.attn_fwd(qkv[:, 0:384], qkv[:, 384:768], qkv[:, 768:1152], B, H, L, L, False)
    oref, _ = _attn_ref(qkv[:, 0:384].contiguous(), qkv[:, 384:768].contiguous(), qkv[:, 768:].contiguous(), B, H, L, L, False)
    assert _rel(o, oref) < 1e-2


@pytest.mark.parametrize("B,H,Lq,Lk,causal", ATTN_SHAPES)
def test_attn_bwd_bf16(dev, tile_rows, B, H, Lq, Lk, causal):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(Lq * 3 + Lk)
    q = (torch.randn(B * Lq, H * 64, generator=g) * 0.35).to(dev).bfloat16()
    k = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    v = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    d_o = torch.randn(B * Lq, H * 64, generator=g).to(dev).bfloat16()
    o, lse = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, causal)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    oref, _ = _attn_ref(qr, kr, vr, B, H, Lq, Lk, causal)
    (oref * d_o.float()).sum().backward()
    # bf16 P/dS operands + bf16 outputs: ~1e-2 relative per tensor
    assert _rel(dq, qr.grad) < 2e-2, _rel(dq, qr.grad)
    assert _rel(dk, kr.grad) < 2e-2, _rel(dk, kr.grad)
    assert _rel(dv, vr.grad) < 2e-2, _rel(dv, vr.grad)


@pytest.mark.parametrize("Lk", [256, 320, 300, 272])
@pytest.mark.parametrize("B,H,Lq,p", [(2, 6, 1024, 0.0), (2, 6, 256, 0.1), (1, 3, 1000, 0.1), (3, 2, 45, 0.0)])
def test_attn_bwd_onepass_matches_autograd_and_the_two_pass_kernels(dev, knobs, B, H, Lq, p, Lk):
    """The one-pass backward (attention_onepass.hip: not causal, 256 keys — the decoder's cross-attention and the encoder's
    self-attention — or, round 6, 256 < Lk <= 320 keys in the 3 + 2 tile form: MR-MT3's own cross-attention over 256 frames + 64
    memory slots, models/t5_segmem_v2_with_prev.py:125-128; 300 and 272 keys exercise the masked key tail, whole idle tiles
    included) against f32 autograd of the same dropped attention — the mask element by element, from the restatement of the
    generator — and against the two-pass kernels on the same inputs (same masks, same bf16 operand roundings: only
    summation orders differ)."""
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(Lq * 3 + int(p * 100))
    q = (torch.randn(B * Lq, H * 64, generator=g) * 0.35).to(dev).bfloat16()
    k = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    v = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    d_o = torch.randn(B * Lq, H * 64, generator=g).to(dev).bfloat16()
    o, lse, o_lo = lib.attn_fwd(q, k, v, B, H, Lq, Lk, False, p=p, seed=31, stream_id=4, want_lo=True)
    res = {}
    for mode in ("1", "0"):
        knobs.set("MRMT3_ATTN_ONEPASS", mode)
        knobs.set("MRMT3_ATTN_ONEPASS_MIN_BH", "1")
        dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        before = lib.dispatch_counts()
        lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, False, p=p, seed=31, stream_id=4, o_lo=o_lo)
        after = lib.dispatch_counts()
        assert after["attn_bwd_onepass"] - before["attn_bwd_onepass"] == int(mode == "1")
        assert after["attn_bwd"] - before["attn_bwd"] == int(mode == "0")
        res[mode] = (dq, dk, dv)
    # f32 autograd of the dropped attention (mask from the restatement of the generator)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    qh = qr.view(B, Lq, H, 64).transpose(1, 2); kh = kr.view(B, Lk, H, 64).transpose(1, 2); vh = vr.view(B, Lk, H, 64).transpose(1, 2)
    pr = torch.softmax(qh @ kh.transpose(2, 3), -1)
    if p > 0:
        from oracle import dropout_ref as dr
        keep, scale = dr.attn_keep_mask(B, H, Lq, Lk, p, 31, 4)
        pr = pr * torch.from_numpy(keep).to(dev) * scale
    ((pr @ vh).transpose(1, 2).reshape(B * Lq, H * 64) * d_o.float()).sum().backward()
    for name, got1, got0, ref in zip(("dq", "dk", "dv"), res["1"], res["0"], (qr.grad, kr.grad, vr.grad)):
        assert torch.isfinite(got1.float()).all(), name
        assert _rel(got1, ref) < 2e-2, (name, _rel(got1, ref))
        assert _rel(got1, got0) < 6e-3, (name, "one-pass vs two-pass", _rel(got1, got0))


def test_attn_online_softmax_rescale_branch(dev):
    """Force the running max to jump at a late key tile (one key spiked against one query)."""
    from mrmt3 import lib
    B, H, L = 1, 1, 512
    q = (torch.randn(L, 64, device=dev) * 0.3).bfloat16()
    k = torch.randn(L, 64, device=dev).bfloat16()
    v = torch.randn(L, 64, device=dev).bfloat16()
    k[450] = (q[17].float() * 40).bfloat16()
    o, lse = lib.attn_fwd(q, k, v, B, H, L, L, False)
    oref, lref = _attn_ref(q, k, v, B, H, L, L, False)
    assert torch.allclose(lse, lref, atol=5e-3, rtol=1e-4) and _rel(o, oref) < 1e-2


@pytest.mark.parametrize("B,H,L,causal", [(1, 2, 256, False), (2, 4, 512, True), (1, 3, 384, True), (64, 6, 1024, True)])
def test_attn_dropout_statistics_and_bwd_mask(dev, tile_rows, B, H, L, causal):
    """Uniform attention (q = k = 0, v = 1): every output is (#kept / #visible) / 0.9.  The causal cases run the
    paired-tile instantiations (an even and an odd number of 128-row tiles, with and without the XCD remap); the last
    case is the decoder self-attention of the benchmark batch at full size."""
    from mrmt3 import lib
    q = torch.zeros(B * L, H * 64, device=dev).bfloat16()          # uniform attention
    k = torch.zeros(B * L, H * 64, device=dev).bfloat16()
    v = torch.ones(B * L, H * 64, device=dev).bfloat16()
    o, lse = lib.attn_fwd(q, k, v, B, H, L, L, causal, p=0.1, seed=99, stream_id=3)
    # each output = (#kept / #visible keys) / 0.9 -> mean 1; rows that see many keys have a small spread
    of = o.float().view(B, L, H, 64)
    assert abs(of[:, L // 2:].mean().item() - 1.0) < 5e-3 and 0.005 < of[:, L // 2:].std().item() < 0.06
    if causal:                               # query 0 sees one key: its output is 0 or 1/0.9 (quantised keep scale 256/230)
        first = of[:, 0, :, 0]
        assert bool(((first == 0) | ((first - 256.0 / 230.0).abs() < 1e-2)).all())
    # dV with dO = 1: dV[k] = sum_q Pd[q,k], and must use the SAME mask as forward:
    d_o = torch.ones_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, L, causal, p=0.1, seed=99, stream_id=3)
    if not causal:
        assert abs(dv.float().mean().item() - 1.0) < 5e-3
    # consistency: sum_k dV[k,d] == sum_q O[q,d] (both equal sum_{q,k} Pd[q,k]) per batch row and head
    s_o = of.sum(1)[..., 0]
    s_dv = dv.float().view(B, L, H, 64).sum(1)[..., 0]
    assert torch.allclose(s_o, s_dv, rtol=3e-3)
    # uniform scores: dS = P * (dP - delta) with dP = mask/0.9 row-constant only up to the mask -> dq, dk stay finite
    assert torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all()


@pytest.mark.parametrize("B,H,Lq,causal", [(2, 3, 200, False), (1, 2, 64, True)])
def test_attn_dropout_mask_equals_the_restatement(dev, tile_rows, B, H, Lq, causal):
    """Uniform scores and V = identity over 64 keys make the output the dropped probability matrix itself:
    O[q, k] = keep[q, k] * scale / (#visible keys).  The kept set must equal oracle/dropout_ref.attn_keep_mask, and
    dV = Pd^T dO must come from the same mask in the backward kernels."""
    from mrmt3 import lib
    from oracle import dropout_ref as dr
    Lk, p, seed, stream = 64, 0.1, 2 ** 35 + 99, 7
    q = torch.zeros(B * Lq, H * 64, device=dev).bfloat16()
    k = torch.zeros(B * Lk, H * 64, device=dev).bfloat16()
    v = torch.eye(64, device=dev).repeat(B, H).bfloat16()                     # [B*64, H*64]: V_h = I for every (b, h)
    o, lse = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream)
    keep, scale = dr.attn_keep_mask(B, H, Lq, Lk, p, seed, stream)
    got = o.float().view(B, Lq, H, 64).permute(0, 2, 1, 3).cpu().numpy()     # [B, H, Lq, Lk]
    qi = torch.arange(Lq).view(Lq, 1).numpy(); ki = torch.arange(Lk).view(1, Lk).numpy()
    visible = (ki <= qi) if causal else (ki >= 0) & (qi >= 0)
    n_vis = visible.sum(1, keepdims=True)
    want = (keep & visible) * (scale / n_vis)
    assert ((got != 0) == (keep & visible)).all()
    assert abs(got - want).max() < 1e-2 * want.max()
    d_o = torch.ones_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream)
    dv_want = want.sum(2)                                                      # [B, H, Lk]: sum_q Pd[q, k]
    dv_got = dv.float().view(B, Lk, H, 64)[..., 0].permute(0, 2, 1).cpu().numpy()
    assert abs(dv_got - dv_want).max() < 2e-2 * max(1.0, dv_want.max())
    # device step counter: forward and backward both follow it
    step = torch.full((1,), 41, device=dev, dtype=torch.int32)
    o2, lse2 = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream, step=step)
    keep2, _ = dr.attn_keep_mask(B, H, Lq, Lk, p, seed, stream, step=41)
    got2 = o2.float().view(B, Lq, H, 64).permute(0, 2, 1, 3).cpu().numpy()
    assert ((got2 != 0) == (keep2 & visible)).all() and (keep2 != keep).mean() > 0.1
    lib.attn_bwd(q, k, v, o2, d_o, lse2, dq, dk, dv, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream, step=step)
    dv_want2 = ((keep2 & visible) * (scale / n_vis)).sum(2)
    dv_got2 = dv.float().view(B, Lk, H, 64)[..., 0].permute(0, 2, 1).cpu().numpy()
    assert abs(dv_got2 - dv_want2).max() < 2e-2 * max(1.0, dv_want2.max())


@pytest.mark.parametrize("causal", [False, True])
def test_attn_dropout_mask_element_by_element_in_every_backward_product(dev, tile_rows, causal):
    """32 queries x 32 keys with K[k] = e_k, Q[q] = e_(32+q), V[k] = e_k and dO = 1: the scores are all zero (uniform
    P), dP[q, k] = 1, so dS[q, k] = P (keep[q, k] * scale - delta_q) takes two values per row and
      dQ[q, k]      = dS[q, k]   (the dQ kernel's mask, query on the lane),
      dK[k, 32 + q] = dS[q, k]   (the dK/dV kernel's mask on dP, key on the lane, words shared through DPP),
      dV[k, k']     = sum_q Pd[q, k] for k' = k... and O[q, k] = Pd[q, k] (forward),
    every one of which must show exactly the kept set of oracle/dropout_ref.attn_keep_mask."""
    from mrmt3 import lib
    from oracle import dropout_ref as dr
    B, H, L, p, seed, stream = 3, 2, 32, 0.1, 2 ** 33 + 5, 11
    eye = torch.eye(64, device=dev)
    k = eye[:32].repeat(B, H).bfloat16()                   # [B*32, H*64]: K[k] = e_k
    q = eye[32:].repeat(B, H).bfloat16()                   # Q[q] = e_(32+q): orthogonal to every key
    v = k.clone()
    o, lse = lib.attn_fwd(q, k, v, B, H, L, L, causal, p=p, seed=seed, stream_id=stream)
    keep, scale = dr.attn_keep_mask(B, H, L, L, p, seed, stream)
    qi, ki = np.arange(L).reshape(L, 1), np.arange(L).reshape(1, L)
    visible = (ki <= qi) if causal else np.ones((L, L), bool)
    kept = keep & visible
    n_vis = visible.sum(1, keepdims=True).astype(np.float64)
    pd = kept * (scale / n_vis)                                               # dropped, rescaled probabilities
    got_o = o.float().view(B, L, H, 64).permute(0, 2, 1, 3).cpu().numpy()[..., :32]
    assert ((got_o != 0) == kept).all()
    d_o = torch.ones_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, L, causal, p=p, seed=seed, stream_id=stream)
    delta = pd.sum(-1, keepdims=True)                                         # rowsum(dO * O) = sum_k Pd[q, k]
    ds_n = visible * (keep * scale - delta)                                   # n_vis * dS[q, k]: O(1) in every row
    got_dq = dq.float().view(B, L, H, 64).permute(0, 2, 1, 3).cpu().numpy()[..., :32]          # [B, H, q, k]
    got_dk = dk.float().view(B, L, H, 64).permute(0, 2, 1, 3).cpu().numpy()[..., 32:]          # [B, H, k, q]
    err_q = np.abs(got_dq * n_vis - ds_n).max()
    err_k = np.abs(got_dk.transpose(0, 1, 3, 2) * n_vis - ds_n).max()
    # kept and dropped elements of a row differ by `scale` = 1.11: a single wrong mask bit is 50x the tolerance
    assert err_q < 2e-2 and err_k < 2e-2, (err_q, err_k)
    got_dv = dv.float().view(B, L, H, 64).permute(0, 2, 1, 3).cpu().numpy()   # dV[k, d] = sum_q Pd[q, k] dO[q, d]
    assert np.abs(got_dv[..., 0] - pd.sum(2)).max() < 2e-2 * pd.sum(2).max()


@pytest.mark.parametrize("B,H,Lq,Lk,causal", [(2, 6, 256, 256, False), (1, 6, 128, 128, True), (1, 6, 100, 320, False)])
def test_attn_f32(dev, B, H, Lq, Lk, causal):
    from mrmt3 import lib
    q = torch.randn(B * Lq, H * 64, device=dev) * 0.35
    k = torch.randn(B * Lk, H * 64, device=dev)
    v = torch.randn(B * Lk, H * 64, device=dev)
    o, lse = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal)
    oref, lref = _attn_ref(q, k, v, B, H, Lq, Lk, causal)
    assert torch.allclose(o, oref, atol=2e-5, rtol=1e-5) and torch.allclose(lse, lref, atol=1e-5)


@pytest.mark.parametrize("B,H,Lq,Lk,causal", [(2, 6, 256, 256, False), (1, 6, 128, 128, True), (1, 3, 100, 320, False)])
def test_attn_bwd_f32_matches_autograd(dev, B, H, Lq, Lk, causal):
    """mrmt3_attn_bwd_f32 (the fp32 training / parity path) against torch autograd of softmax(q k^T) v in f32."""
    from mrmt3 import lib
    q = (torch.randn(B * Lq, H * 64, device=dev) * 0.35).requires_grad_(True)
    k = torch.randn(B * Lk, H * 64, device=dev).requires_grad_(True)
    v = torch.randn(B * Lk, H * 64, device=dev).requires_grad_(True)
    d_o = torch.randn(B * Lq, H * 64, device=dev)
    oref, _ = _attn_ref(q, k, v, B, H, Lq, Lk, causal)
    oref.backward(d_o)
    o, lse = lib.attn_fwd(q.detach(), k.detach(), v.detach(), B, H, Lq, Lk, causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q.detach(), k.detach(), v.detach(), o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, causal)
    for got, ref, name in ((dq, q.grad, "dq"), (dk, k.grad, "dk"), (dv, v.grad, "dv")):
        assert _rel(got, ref) < 2e-5, (name, _rel(got, ref))


def test_attn_f32_dropout_is_the_bf16_kernels_mask_and_consistent_forward_to_backward(dev):
    """The f32 attention kernels draw the mask of the bf16 ones (oracle/dropout_ref.attn_keep_mask): uniform scores and
    V = I expose it in the forward output; dV = Pd^T dO must follow the same mask."""
    from mrmt3 import lib
    from oracle import dropout_ref as dr
    B, H, Lq, Lk, p, seed, stream = 2, 2, 48, 64, 0.1, 77, 3
    q = torch.zeros(B * Lq, H * 64, device=dev)
    k = torch.zeros(B * Lk, H * 64, device=dev)
    v = torch.eye(64, device=dev).repeat(B, H)
    o, lse = lib.attn_fwd(q, k, v, B, H, Lq, Lk, False, p=p, seed=seed, stream_id=stream)
    keep, scale = dr.attn_keep_mask(B, H, Lq, Lk, p, seed, stream)
    got = o.view(B, Lq, H, 64).permute(0, 2, 1, 3).cpu().numpy()
    assert ((got != 0) == keep).all() and abs(got.max() - scale / Lk) < 1e-6
    d_o = torch.ones_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, False, p=p, seed=seed, stream_id=stream)
    want = (keep * (scale / Lk)).sum(2)                                       # [B, H, Lk]
    assert np.abs(dv.view(B, Lk, H, 64)[..., 0].permute(0, 2, 1).cpu().numpy() - want).max() < 1e-5


def _t5_relative_bias(H, Lq, Lk, dev, seed=5):
    """A stock-T5 style position bias [H, Lq, Lk]: a learned table indexed by the bucketed relative position (what HF
    T5Attention.compute_bias builds; MR-MT3 replaces it by zeros, models/t5.py:487-490).  Returns (table, buckets)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    table = torch.randn(32, H, generator=g).to(dev).requires_grad_(True)
    rel = torch.arange(Lk)[None, :] - torch.arange(Lq)[:, None]
    buckets = (rel.clamp(-15, 16) + 15).to(dev)                               # 32 buckets
    return table, buckets


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Lq,Lk,causal,shared", [(2, 6, 128, 128, True, True), (3, 2, 100, 320, False, True),
                                                     (2, 3, 64, 72, False, False), (1, 1, 1, 1, False, True)])
def test_attn_with_additive_bias_matches_autograd_including_the_bias_gradient(dev, dt, B, H, Lq, Lk, causal, shared):
    """mrmt3_attn_fwd_bias / mrmt3_attn_bwd_bias (SURVEY §8b `attn(q, k, v, bias_or_null)`, HF T5Attention's
    `scores += position_bias`, models/t5.py:636-648) against torch autograd in f32: a relative-position bias shared by the
    batch (its gradient sums over the batch and flows on into the bucket table) and a per-sequence bias that also
    masks padded keys with -inf."""
    from mrmt3 import lib
    q = (torch.randn(B * Lq, H * 64, device=dev) * 0.35).to(dt).requires_grad_(True)
    k = torch.randn(B * Lk, H * 64, device=dev).to(dt).requires_grad_(True)
    v = torch.randn(B * Lk, H * 64, device=dev).to(dt).requires_grad_(True)
    d_o = torch.randn(B * Lq, H * 64, device=dev).to(dt)
    table, buckets = _t5_relative_bias(H, Lq, Lk, dev)
    if shared:
        bias = table[buckets].permute(2, 0, 1).contiguous()                    # [H, Lq, Lk]
        full = bias[None]
    else:
        pad = torch.zeros(B, 1, 1, Lk, device=dev)
        for b in range(B):
            pad[b, ..., Lk - 5 * (b + 1):] = float("-inf")                     # the last keys of sequence b are padding
        bias = (table[buckets].permute(2, 0, 1)[None] + pad).contiguous()      # [B, H, Lq, Lk]
        full = bias
    bias.retain_grad()
    qh = q.float().view(B, Lq, H, 64).transpose(1, 2)
    kh = k.float().view(B, Lk, H, 64).transpose(1, 2)
    vh = v.float().view(B, Lk, H, 64).transpose(1, 2)
    sc = qh @ kh.transpose(2, 3) + full
    if causal:
        i = torch.arange(Lq, device=dev)[:, None]
        j = torch.arange(Lk, device=dev)[None, :]
        sc = sc.masked_fill(j > i, float("-inf"))
    oref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B * Lq, H * 64)
    lref = torch.logsumexp(sc, -1)
    oref.backward(d_o.float())

    bd = bias.detach()
    o, lse = lib.attn_fwd_bias(q.detach(), k.detach(), v.detach(), bd, B, H, Lq, Lk, causal)
    tol = 2e-5 if dt == torch.float32 else 6e-3
    assert o.dtype == dt and _rel(o, oref) < tol and torch.allclose(lse, lref, atol=1e-5 if dt == torch.float32 else 1e-4)
    dq, dk, dv, dbias = lib.attn_bwd_bias(q.detach(), k.detach(), v.detach(), o, d_o, lse, bd, B, H, Lq, Lk, causal)
    for got, ref, name in ((dq, q.grad, "dq"), (dk, k.grad, "dk"), (dv, v.grad, "dv"), (dbias, bias.grad, "dbias")):
        # (one key: dS = P (dP - delta) is exactly 0 in autograd and rounding noise here — absolute bound)
        assert got.shape == ref.shape and (_rel(got, ref) < 2 * tol or (got.float() - ref).abs().max() < 1e-5), (name, _rel(got, ref))
    if causal:
        assert (dbias[:, 0, 1:] == 0).all()                                    # masked keys get a zero, not garbage
    # the bias gradient flows on into the learned table exactly as autograd's does
    tg = torch.zeros_like(table).index_put_((buckets.reshape(-1),),
                                            (dbias if shared else dbias.sum(0)).permute(1, 2, 0).reshape(-1, H), accumulate=True)
    assert _rel(tg, table.grad) < 2 * tol or (tg - table.grad).abs().max() < 1e-5
    # no bias at all == the bias-free entry points, bit for bit (f32: same kernel; bf16: general kernel vs MFMA kernel, close)
    o0, l0 = lib.attn_fwd_bias(q.detach(), k.detach(), v.detach(), None, B, H, Lq, Lk, causal)
    o1, l1 = lib.attn_fwd(q.detach(), k.detach(), v.detach(), B, H, Lq, Lk, causal)
    if dt == torch.float32:
        assert torch.equal(o0, o1) and torch.equal(l0, l1)
    else:
        assert _rel(o0, o1) < 1e-2


def test_attn_bias_dropout_draws_the_same_mask_and_rejects_a_bad_bias(dev):
    from mrmt3 import lib
    from oracle import dropout_ref as dr
    B, H, Lq, Lk, p, seed, stream = 2, 2, 48, 64, 0.1, 77, 3
    q = torch.zeros(B * Lq, H * 64, device=dev)
    k = torch.zeros(B * Lk, H * 64, device=dev)
    v = torch.eye(64, device=dev).repeat(B, H)
    bias = torch.zeros(H, Lq, Lk, device=dev)
    o, lse = lib.attn_fwd_bias(q, k, v, bias, B, H, Lq, Lk, False, p=p, seed=seed, stream_id=stream)
    keep, scale = dr.attn_keep_mask(B, H, Lq, Lk, p, seed, stream)
    got = o.view(B, Lq, H, 64).permute(0, 2, 1, 3).cpu().numpy()
    assert ((got != 0) == keep).all() and abs(got.max() - scale / Lk) < 1e-6
    with pytest.raises(AssertionError):
        lib.attn_fwd_bias(q, k, v, torch.zeros(H, Lq, Lk + 1, device=dev), B, H, Lq, Lk, False)
    L = lib.load()
    rc = L.mrmt3_attn_fwd_bias(lib._p(q), q.stride(0), lib._p(k), k.stride(0), lib._p(v), v.stride(0), lib._p(bias), 17,
                               lib._p(o), o.stride(0), lib._p(lse), B, H, Lq, Lk, 0, 0, 0.0, 0, None, 0, lib._stream())
    assert rc != 0 and b"bias batch stride" in L.mrmt3_last_error()


@pytest.mark.parametrize("M,N1,N2", [(512, 512, 384), (1000, 384, 512), (130, 1536, 512), (64, 70, 36)])
def test_gemm_tn_f32(dev, M, N1, N2):
    from mrmt3 import lib
    a = torch.randn(M, N1, device=dev)
    b = torch.randn(M, N2, device=dev)
    c0 = torch.randn(N1, N2, device=dev)
    c = c0.clone()
    lib.gemm_tn_f32(a, b, c, accumulate=True)
    ref = c0.double() + a.double().t() @ b.double()
    assert _rel(c, ref.float()) < 2e-6
    lib.gemm_tn_f32(a[:, :N1 - 2], b[:, 1:], c[:N1 - 2, 1:], accumulate=False)         # strided views, ragged tile edges
    assert _rel(c[:N1 - 2, 1:], (a[:, :N1 - 2].double().t() @ b[:, 1:].double()).float()) < 2e-6
    assert lib.dispatch_counts()["tn_f32"] >= 2


def test_geglu_bwd_and_dropmask_cast_f32(dev):
    from mrmt3 import lib
    h = torch.randn(300, 2048, device=dev).requires_grad_(True)
    dg = torch.randn(300, 1024, device=dev)
    (_gelu_new(h[:, :1024]) * h[:, 1024:] * dg).sum().backward()
    dh = lib.geglu_bwd(h.detach(), dg)
    assert dh.dtype == torch.float32 and _rel(dh, h.grad) < 2e-6
    # with dropout: the f32 forward's mask
    g = lib.geglu_fwd(h.detach(), p=0.1, seed=5, stream_id=9)
    dh2 = lib.geglu_bwd(h.detach(), dg, p=0.1, seed=5, stream_id=9)
    kept = (g != 0).float()
    h2 = h.detach().clone().requires_grad_(True)
    (_gelu_new(h2[:, :1024]) * h2[:, 1024:] * dg * kept / 0.9).sum().backward()
    assert _rel(dh2, h2.grad) < 1e-3
    x = torch.randn(64, 512, device=dev)
    a = lib.dropmask_cast(x, p=0.1, seed=1, stream_id=2, out_dtype=torch.float32)
    b = lib.dropmask_cast(x, p=0.1, seed=1, stream_id=2)
    assert a.dtype == torch.float32 and torch.equal(a.bfloat16(), b) and torch.equal(lib.dropmask_cast(x, out_dtype=torch.float32), x)


# ---- gated GELU, embedding, CE, AdamW ------------------------------------------------------------------

def _gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def test_geglu(dev):
    from mrmt3 import lib
    h = torch.randn(777, 2048, device=dev)
    g = lib.geglu_fwd(h)
    assert torch.allclose(g, _gelu_new(h[:, :1024]) * h[:, 1024:], atol=1e-6, rtol=1e-5)
    hb = h.bfloat16()
    gb = lib.geglu_fwd(hb)
    hr = hb.float().requires_grad_(True)
    gr = _gelu_new(hr[:, :1024]) * hr[:, 1024:]
    assert _rel(gb, gr) < 4e-3
    dg = torch.randn(777, 1024, device=dev).bfloat16()
    (gr * dg.float()).sum().backward()
    dh = lib.geglu_bwd(hb, dg)
    assert _rel(dh, hr.grad) < 4e-3


def test_embed_shift_right_and_scatter(dev):
    from mrmt3 import lib
    from mrmt3.synthetic import sinusoid_table, synth_labels
    B, L, d, V = 3, 64, 512, 1536
    table = torch.randn(V, d, device=dev)
    pos = sinusoid_table(128, d).to(dev)
    labels = torch.from_numpy(synth_labels(B, L, full=False, mean_len=20)).to(dev)
    x = lib.embed_fwd(labels, table, pos, L, shift=True)
    ids = torch.cat([torch.zeros(B, 1, dtype=torch.long, device=dev), labels[:, :-1]], 1)
    ids = ids.masked_fill(ids == -100, 0)
    ref = table[ids] + pos[None, :L]
    assert torch.equal(x.view(B, L, d), ref)                    # gather + one fp32 add: bit-exact
    dx = torch.randn(B * L, d, device=dev)
    dt = torch.zeros(V, d, device=dev)
    lib.embed_bwd(labels, dx, dt, L, shift=True)
    dref = torch.zeros(V, d, device=dev).index_add_(0, ids.view(-1), dx)
    assert torch.allclose(dt, dref, atol=1e-5)
    x2 = lib.embed_fwd(ids, table, pos, L, shift=False, pos_offset=3)
    assert torch.equal(x2.view(B, L, d), table[ids] + pos[None, 3:3 + L])
    src = torch.randn(B * L, d, device=dev)
    assert torch.equal(lib.addpos_fwd(src, pos, L).view(B, L, d), src.view(B, L, d) + pos[None, :L])


@pytest.mark.parametrize("skew", ["uniform", "padded", "one_id"])
def test_embed_bwd_sorted_sum_is_exact_and_reproducible(dev, skew):
    """The embedding gradient is a sorted segmented sum (no atomics): equals a float64 index_add to f32 rounding,
    accumulates into what the table already holds, is bitwise identical across runs, for uniform ids, for
    Slakh-shaped rows (most positions are padding -> id 0) and when every row carries the same id."""
    from mrmt3 import lib
    B, L, d, V = 16, 1024, 512, 1536
    g = torch.Generator().manual_seed(5)
    if skew == "uniform":
        labels = torch.randint(3, 1391, (B, L), generator=g)
    elif skew == "padded":
        labels = torch.full((B, L), -100, dtype=torch.long)
        for b in range(B):
            n = int(torch.randint(50, 400, (1,), generator=g))
            labels[b, :n] = torch.randint(1129, 1140, (n,), generator=g)       # a handful of hot ids
            labels[b, n] = 1
    else:
        labels = torch.full((B, L), 7, dtype=torch.long)
    labels = labels.to(dev)
    dx = torch.randn(B * L, d, generator=g).to(dev)
    ids = torch.cat([torch.zeros(B, 1, dtype=torch.long, device=dev), labels[:, :-1]], 1)
    ids = ids.masked_fill(ids == -100, 0).view(-1)
    start = torch.randn(V, d, generator=g).to(dev)
    want = start.double().index_add_(0, ids, dx.double())
    outs = []
    for _ in range(2):
        dt = start.clone()
        lib.embed_bwd(labels, dx, dt, L, shift=True)
        outs.append(dt)
    assert torch.equal(outs[0], outs[1])
    counts = torch.bincount(ids, minlength=V).double().clamp(min=1)
    err = (outs[0].double() - want).abs().max(dim=1).values
    assert (err <= 4e-6 * counts.sqrt() * 8 + 1e-5).all(), err.max().item()
    # dropout: the kept elements are those of the forward mask stream (same seed / stream id), scaled by 1/(1-p)
    dt = torch.zeros(V, d, device=dev)
    lib.embed_bwd(labels, torch.ones(B * L, d, device=dev), dt, L, shift=True, p=0.25, seed=9, stream_id=4)
    kept = dt.sum().item() / (B * L * d) * 0.75
    assert abs(kept - 0.75) < 5e-3


@pytest.mark.parametrize("weighted", [False, True])
def test_cross_entropy(dev, weighted):
    from mrmt3 import lib
    from mrmt3.synthetic import synth_labels
    from oracle import t5_ref
    rows, V = 2048, 1536
    logits = torch.randn(rows, V, device=dev) * 2
    tg = torch.from_numpy(synth_labels(2, 1024, full=False)).to(dev).view(-1)
    if weighted:
        tg = tg.masked_fill(tg == -100, 5)   # the reference's weighted loss cannot take -100 rows... keep some
        tg[::7] = -100
    loss, dl = lib.cross_entropy(logits, tg, grad_dtype=torch.float32, weighted=weighted)
    lr = logits.double().cpu().requires_grad_(True)
    ref = (t5_ref.weighted_ce_loss if weighted else t5_ref.ce_loss)(lr[None], tg.cpu()[None])
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert torch.allclose(dl.cpu().double(), lr.grad, atol=1e-8, rtol=1e-4)


def test_adamw_matches_torch(dev):
    from mrmt3 import lib
    n = 4096 * 3
    p0 = torch.randn(n, device=dev)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=2e-4)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lr = torch.tensor([2e-4], device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    shadow = torch.empty(n, device=dev, dtype=torch.bfloat16)
    for i in range(5):
        g = torch.randn(n, device=dev)
        ref_p.grad = g.clone()
        opt.step()
        lib.adamw_step(p, g, m, v, lr, step, shadow=shadow)
    assert step.item() == 5
    assert torch.allclose(p, ref_p.data, atol=1e-7, rtol=1e-6)
    assert torch.equal(shadow, p.bfloat16())


def test_transpose_cast(dev):
    from mrmt3 import lib
    w = torch.randn(384, 512, device=dev)
    out = torch.empty(512, 384, device=dev, dtype=torch.bfloat16)
    lib.transpose(w, out)
    assert torch.equal(out, w.t().contiguous().bfloat16())
    c = torch.empty(384 * 512, device=dev, dtype=torch.bfloat16)
    lib.cast(w.view(-1), c)
    assert torch.equal(c, w.view(-1).bfloat16())


def test_transpose_batched_even_and_odd_shapes(dev):
    """One launch, several matrices in flat buffers: pair-vectorised path (even dims / offsets) and the scalar
    fallback (odd dims or offsets), tiles that overhang the matrix."""
    import numpy as np
    from mrmt3 import lib
    shapes = [(384, 512), (70, 130), (33, 7), (64, 64), (5, 1000), (136, 72), (2048, 512), (72, 200)]   # (multiples of 8: the 16-byte path inside, the scalar path on overhanging tiles)
    src_parts, recs, starts, tot, so, do = [], [], [], 0, 0, 0
    for r, c in shapes:
        src_parts.append(torch.randn(r * c, device=dev).bfloat16())
        recs.append((so, do, r, c))
        starts.append(tot)
        tot += ((r + 63) // 64) * ((c + 63) // 64)
        so += r * c
        do += r * c
    src = torch.cat(src_parts)
    dst = torch.zeros_like(src)
    tab = np.zeros(len(recs), dtype=[("src", "<i8"), ("dst", "<i8"), ("rows", "<i4"), ("cols", "<i4")])
    for i, t in enumerate(recs):
        tab[i] = t
    lib.transpose_batched(src, dst, torch.from_numpy(tab.view(np.uint8).copy()).to(dev),
                          torch.tensor(starts, dtype=torch.int32, device=dev), len(recs), tot)
    for (so_, do_, r, c) in recs:
        assert torch.equal(dst[do_:do_ + r * c].view(c, r), src[so_:so_ + r * c].view(r, c).t())


def test_errors_are_reported_not_fatal(dev):
    from mrmt3 import lib
    a = torch.randn(64, 40, device=dev).bfloat16()   # K*2 not a multiple of 64
    with pytest.raises(RuntimeError, match="multiple of 64"):
        lib.gemm_nt(a, a)
    with pytest.raises(RuntimeError, match="device tensors"):
        lib.gemm_nt(torch.zeros(128, 64).bfloat16(), torch.zeros(128, 64).bfloat16())


def test_attn_bwd_delta_from_hi_lo_output_keeps_the_common_mode_cancellation(dev):
    """Value rows with a large common component (V_j = vbar + small): dS_ij = P_ij dO_i.(V_j - O_i) cancels vbar
    exactly only if delta = dO.O sees O beyond bf16.  With the forward's low half (o_lo) handed to the backward the
    dq / dk error against an f64 reference drops by several times; without it the old behaviour remains."""
    from mrmt3 import lib
    torch.manual_seed(0)
    B, H, Lq, Lk = 2, 6, 256, 256
    q = (torch.randn(B * Lq, H * 64, device=dev) * 0.3).bfloat16()
    k = (torch.randn(B * Lk, H * 64, device=dev) * 0.3).bfloat16()
    v = (torch.randn(1, H * 64, device=dev) + 0.1 * torch.randn(B * Lk, H * 64, device=dev)).bfloat16()
    d_o = (torch.randn(B * Lq, H * 64, device=dev) * 1e-2).bfloat16()
    o, lse, o_lo = lib.attn_fwd(q, k, v, B, H, Lq, Lk, False, want_lo=True)
    o2, _ = lib.attn_fwd(q, k, v, B, H, Lq, Lk, False)
    assert torch.equal(o, o2) and o_lo is not None

    def heads(t, L):
        return t.double().view(B, L, H, 64).permute(0, 2, 1, 3)
    qd, kd, vd, dod = heads(q, Lq), heads(k, Lk), heads(v, Lk), heads(d_o, Lq)
    P = torch.softmax(qd @ kd.transpose(-1, -2), -1)
    O = P @ vd
    # the low half is the rounding residual of the high half (|lo| <= half an ulp of hi) and moves the sum towards O
    assert (o_lo.float().abs() <= o.float().abs() * 2.0 ** -8 + 1e-30).all()
    assert (heads(o, Lq) + heads(o_lo, Lq) - O).norm() < 0.7 * (heads(o, Lq) - O).norm()
    dP = dod @ vd.transpose(-1, -2)
    dS = P * (dP - (P * dP).sum(-1, keepdim=True))
    dq_ref, dk_ref = dS @ kd, dS.transpose(-1, -2) @ qd
    err = {}
    for name, lo in (("hi_only", None), ("hi_lo", o_lo)):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, False, o_lo=lo)
        err[name] = (((heads(dq, Lq) - dq_ref).norm() / dq_ref.norm()).item(),
                     ((heads(dk, Lk) - dk_ref).norm() / dk_ref.norm()).item())
    print("attention backward, common-mode V: rel err (dq, dk)", err)
    assert err["hi_lo"][0] < 6e-3 and err["hi_lo"][1] < 6e-3
    assert err["hi_only"][0] > 2.5 * err["hi_lo"][0] and err["hi_only"][1] > 2.5 * err["hi_lo"][1]


def test_gemm_tn_deferred_batch_is_bitwise_the_immediate_form(dev):
    """lib.TnBatch: the weight-gradient GEMMs of a step leave their split-K slabs per site and ONE launch sums them all
    (`mrmt3_tn_reduce_sites`).  Same per-element summation order as `mrmt3_gemm_tn`: bit-identical, accumulate
    included, tables reused across flushes."""
    from mrmt3 import lib
    torch.manual_seed(1)
    shapes = [(4096, 512, 384), (2048 + 64, 1152, 512), (4096, 2048, 512), (1024, 384, 512), (8192, 512, 1024)]
    ops = [((torch.randn(M, N1, device=dev) * 0.1).bfloat16(), (torch.randn(M, N2, device=dev) * 0.1).bfloat16())
           for M, N1, N2 in shapes]
    want = [torch.randn(N1, N2, device=dev) for _, N1, N2 in shapes]
    got = [w.clone() for w in want]
    for (a, b), w in zip(ops, want):
        lib.gemm_tn(a, b, w, accumulate=True)
    batch = lib.TnBatch()
    for rep in range(2):
        for (a, b), g in zip(ops, got):
            lib.gemm_tn(a, b, g, accumulate=True, defer=batch)
        batch.flush()
        batch.flush()                       # empty flush is a no-op
        if rep == 0:
            for g, w in zip(got, want):
                assert torch.equal(g, w)
            for (a, b), w in zip(ops, want):
                lib.gemm_tn(a, b, w, accumulate=True)
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    assert len(batch._tables) == 1
    # strided C (a row block of a fused weight gradient) and accumulate=False
    big = torch.zeros(1152, 512, device=dev)
    ref = torch.zeros(384, 512, device=dev)
    a, b = ops[3]
    lib.gemm_tn(a, b, ref, accumulate=False)
    lib.gemm_tn(a, b, big[384:768], accumulate=False, defer=batch)
    batch.flush()
    assert torch.equal(big[384:768], ref) and big[:384].abs().max() == 0 and big[768:].abs().max() == 0


@pytest.mark.parametrize("M,N,K,out,acc", [
    (65536, 512, 384, "bf16", False), (16384, 2048, 512, "bf16", False), (65536, 512, 2048, "bf16", False),
    (65536, 1536, 512, "f32", False), (16384, 512, 768, "f32", True),
    (4096 + 200, 1152, 512, "bf16", False),      # ragged M (zero-filled rows, masked stores) + half-overlapping last column tile
    (8192, 384, 128, "bf16", False),             # two K steps per tile: first, second and last K step coincide
    (4096, 768, 256, "f32", False),
    # 12 segments per GPU (the reference's own batch): 12288 decoder rows, 3072 encoder rows, less than one wave of tiles
    (3072, 512, 1024, "bf16", False), (3072, 2048, 512, "bf16", False), (12288, 384, 512, "bf16", False),
    (12288, 512, 1536, "f32", False), (2048 + 72, 512, 384, "bf16", False)])
def test_gemm_nt8_pingpong_kernel_against_f32_and_the_first_kernel(dev, knobs, M, N, K, out, acc):
    """csrc/gemm8.hip (ping-pong phases, LDS-DMA two K steps ahead, epilogue spread over four phases with counted
    waits) on every admissible shape class: against an f32 torch product on ALL rows, and against gemm.hip's kernel
    (same MFMA, same k order: expected bit-identical for f32 output)."""
    from mrmt3 import lib
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    dt = torch.bfloat16 if out == "bf16" else torch.float32
    base = torch.randn(M, N, device=dev).to(dt) if acc else torch.zeros(M, N, device=dev, dtype=dt)
    res = {}
    for which in ("0", "1"):
        knobs.set("MRMT3_GEMM8", which)
        knobs.set("MRMT3_GEMM8_ALL", "1")
        c = base.clone()
        for _ in range(2 if acc else 1):
            lib.gemm_nt(a, b, out=c, accumulate=acc)
        res[which] = c
    ref = a.float() @ b.float().t()
    if acc:
        ref = base.float() + 2 * ref
    err = (res["1"].float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (6e-3 if out == "bf16" else 2e-6), err
    if out == "f32":
        assert torch.equal(res["0"], res["1"])
    else:
        assert (res["0"].float() - res["1"].float()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    # operands and output as column slices of wider buffers (q | k | v views of the fused projection)
    knobs.set("MRMT3_GEMM8", "1")
    wide_a = torch.randn(M, K + 128, device=dev).bfloat16()
    wide_c = torch.zeros(M, N + 256, device=dev, dtype=dt)
    lib.gemm_nt(wide_a[:, 128:], b, out=wide_c[:, 256:])
    ref2 = wide_a[:, 128:].float() @ b.float().t()
    assert (wide_c[:, 256:].float() - ref2).abs().max().item() / ref2.abs().max().item() < (6e-3 if out == "bf16" else 2e-6)
    assert wide_c[:, :256].abs().max().item() == 0


def test_gemm8_dispatch_rule_takes_whole_waves_of_tiles(dev, knobs):
    """Which NT shapes go to the ping-pong kernel by default (no tuning switches): at most one wave of workgroups, or a
    last wave >= 80 % full; never the accumulate form.  (Cold A/B: profiles/r03_gemm_ab_*cold.txt.)"""
    from mrmt3 import lib
    knobs.unset("MRMT3_GEMM8_ALL")
    knobs.unset("MRMT3_GEMM8")
    knobs.unset("MRMT3_GEMM8_MIN_M")
    cases = [(65536, 1152, 512, True), (65536, 384, 512, True), (16384, 1152, 512, False), (12288, 512, 384, True),
             (12288, 1024, 512, False), (12288, 1536, 512, False), (3072, 512, 2048, True), (3072, 1152, 512, True),
             (1024, 512, 512, False), (65536, 512, 1024, True)]
    for M, N, K, want in cases:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        c0 = lib.dispatch_counts()
        c = lib.gemm_nt(a, b)
        c1 = lib.dispatch_counts()
        # (short inputs with K >= 2048 run the same kernel split over K: its own counter)
        took = (c1["gemm_nt8"] - c0["gemm_nt8"]) + (c1["gemm_nt_splitk"] - c0["gemm_nt_splitk"])
        assert (took == 1) == want, (M, N, K, want)
        assert (c1["gemm_nt_splitk"] - c0["gemm_nt_splitk"] == 1) == (M <= 4096 and K >= 2048), (M, N, K)
        rows = torch.randint(0, M, (64,), device=dev)
        ref = a[rows].float() @ b.float().t()
        assert (c[rows].float() - ref).abs().max().item() / ref.abs().max().item() < 6e-3
    acc = torch.zeros(65536, 512, device=dev)
    before = lib.dispatch_counts()["gemm_nt8"]
    lib.gemm_nt(torch.randn(65536, 384, device=dev).bfloat16(), torch.randn(512, 384, device=dev).bfloat16(), out=acc, accumulate=True)
    assert lib.dispatch_counts()["gemm_nt8"] == before


def test_gemm8_and_tn8_race_screen_bitwise_repeatable_under_load(dev, knobs):
    """The ping-pong kernels order every LDS hand-off by counted vmcnt + barriers (no fences): a misplaced wait shows up
    as rare wrong tiles that come and go with timing.  Screen: 150 back-to-back launches per shape, alternating with a
    bandwidth-heavy kernel (different timing every time), every result bit-identical to the first and equal to an f32
    reference."""
    from mrmt3 import lib
    knobs.set("MRMT3_GEMM8_ALL", "1")
    knobs.set("MRMT3_TN8_ALL", "1")
    torch.manual_seed(3)
    noise = torch.empty(64 << 20, device=dev)
    for M, N, K in ((16384, 512, 384), (65536, 1152, 512), (32768, 2048, 128)):
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        first = lib.gemm_nt(a, b, out_dtype=torch.float32).clone()
        ref = a.float() @ b.float().t()
        assert (first - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-6
        out = torch.empty_like(first)
        bad = 0
        for it in range(150):
            if it % 3 == 0:
                noise.fill_(float(it))
            lib.gemm_nt(a, b, out=out)
            bad += int(not torch.equal(out, first))
        assert bad == 0, (M, N, K, bad)
    for M, N1, N2 in ((65536, 512, 1024), (32768, 1152, 512)):
        a = (torch.randn(M, N1, device=dev) * 0.1).bfloat16()
        b = (torch.randn(M, N2, device=dev) * 0.1).bfloat16()
        first = torch.zeros(N1, N2, device=dev)
        lib.gemm_tn(a, b, first)
        ref = a.float().t() @ b.float()
        assert (first - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
        out = torch.empty_like(first)
        bad = 0
        for it in range(100):
            if it % 3 == 0:
                noise.fill_(float(it))
            lib.gemm_tn(a, b, out)
            bad += int(not torch.equal(out, first))
        assert bad == 0, (M, N1, N2, bad)


@pytest.mark.parametrize("weighted", [False, True])
def test_lmhead_ce_fused_chunks_equal_gemm_then_ce(dev, weighted):
    """mrmt3_lmhead_ce_fwd_bwd: lm_head + CE over row chunks with the f32 logits only in a chunk-sized
    workspace == mrmt3_gemm_nt followed by mrmt3_ce_fwd_bwd on the whole [rows, V] tensor: dlogits bit-identical, loss equal
    up to the order of its float-atomic accumulation; also against the f64 torch loss."""
    from mrmt3 import lib
    from mrmt3.synthetic import synth_labels
    torch.manual_seed(7)
    rows, d, V = 5 * 1024 + 256, 512, 1536
    dec = torch.randn(rows, d, device=dev).bfloat16()
    w = (torch.randn(V, d, device=dev) * 0.05).bfloat16()
    tg = torch.from_numpy(synth_labels(rows // 256, 256, full=False, seed=3, mean_len=120)).to(dev).reshape(-1)
    if weighted:
        tg[::7] = 1200                                      # instrument tokens (weight 3 / count 2)
    logits = lib.gemm_nt(dec, w, out_dtype=torch.float32)
    loss_a, dl_a = lib.cross_entropy(logits, tg, weighted=weighted)
    for chunk in (1024, 4096, 1 << 20):
        loss_b, dl_b = lib.lmhead_cross_entropy(dec, w, tg, weighted=weighted, chunk_rows=chunk)
        assert torch.equal(dl_a, dl_b)
        assert abs(loss_a.item() - loss_b.item()) < 2e-6 * max(1.0, abs(loss_a.item()))
    loss_c, none = lib.lmhead_cross_entropy(dec, w, tg, want_grad=False, weighted=weighted)
    assert none is None and abs(loss_c.item() - loss_a.item()) < 2e-6 * max(1.0, abs(loss_a.item()))
    if not weighted:
        ref = torch.nn.functional.cross_entropy(logits.double(), tg, ignore_index=-100).item()
        assert abs(loss_b.item() - ref) < 2e-5


@pytest.mark.gpu
def test_gemm_nt_geglu_fused_equals_the_two_kernels_bitwise(dev, knobs):
    """K2 + K7 in one launch (the wi projection with the gated GELU and its dropout in the GEMM epilogue): h and g are
    bit-identical to mrmt3_gemm_nt followed by mrmt3_geglu_fwd — same bf16 rounding of h before the activation, same
    counter-based mask (keyed on the flat index of g, salted by the device step) — on 256- and 128-row tiles, ragged
    row counts and the shapes that fall back to the two kernels; and g matches the f32 formula
    (models/t5.py T5DenseGatedGeluDense.forward: gelu_new(wi_0 x) * wi_1 x)."""
    from mrmt3 import lib
    torch.manual_seed(11)
    step = torch.tensor([7], device=dev, dtype=torch.int32)
    for rows, dff, K in ((32768, 1024, 512), (65536 + 200, 1024, 512), (4096, 1024, 512), (8192 + 72, 512, 256),
                         (1024, 1024, 512), (4096, 320, 512),
                         (3072, 1024, 512), (2048 + 72, 1024, 512)):       # round 6: fused from 2048 rows (12 segments: 3072 encoder rows)
        x = torch.randn(rows, K, device=dev).bfloat16()
        wi = (torch.randn(2 * dff, K, device=dev) * 0.06).bfloat16()
        for p in (0.0, 0.1):
            knobs.set("MRMT3_GEGLU_FUSED", "0")
            h0, g0 = lib.gemm_nt_geglu(x, wi, p=p, seed=1234, stream_id=5, step=step if p else None)
            h1 = lib.gemm_nt(x, wi)
            g1 = lib.geglu_fwd(h1, p=p, seed=1234, stream_id=5, step=step if p else None)
            assert torch.equal(h0, h1) and torch.equal(g0, g1)
            knobs.set("MRMT3_GEGLU_FUSED", "1")
            h, g = lib.gemm_nt_geglu(x, wi, p=p, seed=1234, stream_id=5, step=step if p else None)
            assert torch.equal(h, h0), (rows, dff, K, p, (h.float() - h0.float()).abs().max().item())
            assert torch.equal(g, g0), (rows, dff, K, p, (g.float() - g0.float()).abs().max().item())
            if p == 0.0:
                hf = x.float() @ wi.float().t()
                ref = torch.nn.functional.gelu(hf[:, :dff], approximate="tanh") * hf[:, dff:]
                err = (g.float() - ref).abs().max().item()
                assert err < 2e-2 * ref.abs().max().item() + 1e-3, (rows, dff, err)
            else:
                kept = (g != 0).float().mean().item()
                assert abs(kept - 0.9) < 5e-3
    # a different step draws a different mask; the same step the same one
    x = torch.randn(8192, 512, device=dev).bfloat16()
    wi = (torch.randn(2048, 512, device=dev) * 0.06).bfloat16()
    ga = lib.gemm_nt_geglu(x, wi, p=0.1, seed=9, stream_id=2, step=step)[1]
    gb = lib.gemm_nt_geglu(x, wi, p=0.1, seed=9, stream_id=2, step=step)[1]
    step.add_(1)
    gc = lib.gemm_nt_geglu(x, wi, p=0.1, seed=9, stream_id=2, step=step)[1]
    assert torch.equal(ga, gb) and not torch.equal(ga, gc)


@pytest.mark.gpu
def test_grouped_weight_gradients_match_f32_and_are_bitwise_repeatable(dev, knobs):
    """mrmt3_tn_group_plan / _run: several weight gradients dW = dY^T X in one MFMA launch + one reduce.  Every shape of
    the training step (ragged 384 / 768 / 1152 with their shifted last tiles, strided operand views, a token count that
    is not a multiple of 128), mixed token counts in one group, accumulate on and off; against an f32 matmul of the same
    bf16 operands, bit-identical when repeated (fixed summation order), and equal to the one-by-one kernels within f32
    summation-order noise.  Reference op: autograd of nn.Linear(bias=False) wrt its weight (models/t5.py:51,72)."""
    from mrmt3 import lib
    torch.manual_seed(5)
    shapes = [(65536, 512, 1024), (65536, 2048, 512), (65536, 512, 384), (65536, 384, 512), (16384, 768, 512),
              (65536, 1152, 512), (16384 + 72, 512, 512), (4096, 1536, 512), (1024, 256, 256)]
    for n_ctas in (None, "24"):           # the real chip, and a small grid (many rounds per workgroup)
        if n_ctas:
            knobs.set("MRMT3_TN_GROUP_CTAS", n_ctas)
            shapes = shapes[4:]
        grp = lib.TnGroup()
        sites = []
        for i, (M, N1, N2) in enumerate(shapes):
            wide = torch.randn(M, N1 + 128, device=dev).mul_(0.1).bfloat16()
            a = wide[:, 64:64 + N1] if i % 2 else wide[:, :N1].contiguous()          # a strided view every other site
            b = torch.randn(M, N2, device=dev).mul_(0.1).bfloat16()
            acc = i % 3 == 0
            out = torch.randn(N1, N2, device=dev) if acc else torch.full((N1, N2), float("nan"), device=dev)
            init = out.clone()
            assert lib.TnGroup.ok(a, b, out)
            sites.append((a, b, out, acc, init))
        for a, b, out, acc, _ in sites:
            grp.add(a, b, out, accumulate=acc)
        grp.flush()
        info = grp.last_info
        assert info.n_items >= info.n_ctas // 2 and info.rounds >= (1 if n_ctas is None else 4), (info.n_items, info.rounds)
        firsts = []
        for a, b, out, acc, init in sites:
            ref = a.float().t() @ b.float()
            if acc:
                ref = ref + init
            err = (out - ref).abs().max().item()
            assert err < 3e-5 * ref.abs().max().item() + 1e-5, (a.shape, b.shape, err)
            firsts.append(out.clone())
            one = torch.zeros_like(out)
            lib.gemm_tn(a, b, one)
            if acc:
                one += init
            assert (out - one).abs().max().item() < 3e-5 * ref.abs().max().item() + 1e-5
        for rep in range(3):               # same plan -> same bits
            for a, b, out, acc, init in sites:
                out.copy_(init)
                grp.add(a, b, out, accumulate=acc)
            grp.flush()
            for (a, b, out, acc, init), f in zip(sites, firsts):
                assert torch.equal(out, f)
    assert not lib.TnGroup.ok(torch.zeros(512, 512, device=dev).bfloat16(), torch.zeros(512, 512, device=dev).bfloat16(),
                              torch.zeros(512, 512, device=dev))         # too few rows: goes through mrmt3_gemm_tn
    assert not lib.TnGroup.ok(torch.zeros(4096, 320, device=dev).bfloat16(), torch.zeros(4096, 512, device=dev).bfloat16(),
                              torch.zeros(320, 512, device=dev))


@pytest.mark.parametrize("causal,Lq,Lk", [(True, 256, 256), (False, 200, 256), (False, 128, 320)])
def test_attn_fwd_row_store_paths_agree(dev, causal, Lq, Lk):
    """The forward kernel writes its output rows 16 bytes per lane (two v_permlane16_swap per chunk pair) when the rows are
    16-byte aligned, 8 bytes per lane otherwise: same bits either way, for O, its low half and the log-sum-exp, with a ragged
    last query tile and with dropout."""
    from mrmt3 import lib
    B, H = 3, 6
    torch.manual_seed(5)
    qkv = torch.randn(B * max(Lq, Lk), 1152, device=dev).bfloat16()
    q, k, v = qkv[:B * Lq, :384], qkv[:B * Lk, 384:768], qkv[:B * Lk, 768:]
    o, lse, o_lo = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=0.1, seed=9, stream_id=2, want_lo=True)
    for ld, col0 in ((388, 0), (392, 4)):           # rows 8 bytes off / aligned stride but an 8-byte offset base
        buf = torch.full((B * Lq, ld), 7.0, device=dev).bfloat16()
        buf_lo = torch.full((B * Lq, ld), 7.0, device=dev).bfloat16()
        o2, lse2, lo2 = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=0.1, seed=9, stream_id=2, want_lo=True,
                                     out=buf[:, col0:col0 + 384], out_lo=buf_lo[:, col0:col0 + 384])
        assert torch.equal(o2, o) and torch.equal(lo2, o_lo) and torch.equal(lse2, lse)
        pad = torch.ones(ld, dtype=torch.bool, device=dev)
        pad[col0:col0 + 384] = False
        assert (buf[:, pad] == 7.0).all() and (buf_lo[:, pad] == 7.0).all()      # nothing written outside the rows
