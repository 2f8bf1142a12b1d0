"""Pins the CPU oracle (oracle/t5_ref.py) against vectors recorded from the reference itself
(tests/golden/make_golden.py, run in the build container through the HF-compat shim)."""
import numpy as np
import pytest
import torch

from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
from oracle import t5_ref

VARIANTS = ["t5", "segmem_v1", "segmem_v2", "segmem_v2_with_prev"]


def _sd(variant):
    w = golden_weights(T5_SMALL, 0 if variant == "t5" else 1)
    return {k: torch.from_numpy(v) for k, v in w.items()}


def _inputs():
    B = 2
    return (torch.from_numpy(synth_mel(B)), torch.from_numpy(synth_labels(B, full=True)),
            torch.from_numpy(synth_labels(B, full=False, seed=777)),
            torch.from_numpy(synth_labels(B, full=False, seed=999)))


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("tag", ["full", "pad"])
def test_logits_and_loss_match_reference(golden, variant, tag):
    torch.set_num_threads(8)
    mel, lab_full, lab_pad, prev = _inputs()
    lab = lab_full if tag == "full" else lab_pad
    with torch.no_grad():
        logits = t5_ref.forward_logits(_sd(variant), T5_SMALL, mel, lab, variant=variant,
                                       targets_prev=prev.clone())
    idx = golden[f"{variant}.{tag}.logit_idx"]
    got = logits.reshape(-1)[idx].numpy()
    np.testing.assert_allclose(got, golden[f"{variant}.{tag}.logit_val"], atol=2e-5, rtol=0)
    loss = t5_ref.ce_loss(logits.double(), lab).item()
    assert abs(loss - float(golden[f"{variant}.{tag}.loss"])) < 1e-5
    agree = (logits.argmax(-1).numpy() == golden[f"{variant}.{tag}.argmax"]).mean()
    assert agree > 0.999


def test_greedy_t5_matches_reference(golden):
    mel = torch.from_numpy(synth_mel(2))
    with torch.no_grad():
        ids = t5_ref.generate_t5(_sd("t5"), T5_SMALL, mel, max_length=32)
    np.testing.assert_array_equal(ids.numpy(), golden["t5.gen32"])


def test_greedy_segmem_v1_matches_reference(golden):
    """T5SegMem.generate (plain batched decode) and generate_2 (memory prepended to the decoder input)."""
    mel = torch.from_numpy(synth_mel(2))
    with torch.no_grad():
        ids = t5_ref.generate_t5(_sd("segmem_v1"), T5_SMALL, mel, max_length=32)
        ids2 = t5_ref.generate_segmem_v1(_sd("segmem_v1"), T5_SMALL, mel, max_length=96)
    np.testing.assert_array_equal(ids.numpy(), golden["segmem_v1.gen32"])
    np.testing.assert_array_equal(ids2.numpy(), golden["segmem_v1.gen2_96"])


@pytest.mark.parametrize("variant", ["segmem_v2", "segmem_v2_with_prev"])
def test_greedy_segmem_matches_reference(golden, variant):
    mel = torch.from_numpy(synth_mel(2))
    with torch.no_grad():
        ids = t5_ref.generate_segmem_v2(_sd(variant), T5_SMALL, mel, max_length=32,
                                        with_prev=variant.endswith("prev"))
    np.testing.assert_array_equal(ids.numpy(), golden[f"{variant}.gen32"])


def test_cosine_schedule_known_answers():
    # utils.py:53-61 — warmup is linear, min_lr floors the multiplier
    f = lambda s: t5_ref.cosine_lambda(s, 100, 1000, min_lr=1e-4)
    assert f(0) == 0.0 and f(50) == 0.5 and f(100) == 1.0
    assert abs(f(550) - 0.5) < 1e-12
    assert f(1000) == 1e-4 and f(999) >= 1e-4


def test_greedy_fixtures_are_not_knife_edge(golden):
    """bit-exact token comparisons only mean something if no recorded argmax sits on a
    near-tie.  Top-2 logit margins of the oracle along the recorded 32-token decodes must stay far above the
    1e-4 logit agreement the fp32 GPU path is held to."""
    mel = torch.from_numpy(synth_mel(2))
    with torch.no_grad():
        ids, margins = t5_ref.generate_t5(_sd("t5"), T5_SMALL, mel, max_length=32, return_margins=True)
    np.testing.assert_array_equal(ids.numpy(), golden["t5.gen32"])
    assert float(margins.min()) > 1e-3, float(margins.min())          # 5x the 2e-4 logit tolerance of the fp32 GPU path
    with torch.no_grad():
        ids2, m2 = t5_ref.generate_segmem_v2(_sd("segmem_v2_with_prev"), T5_SMALL, mel, max_length=32, with_prev=True,
                                             return_margins=True)
    np.testing.assert_array_equal(ids2.numpy(), golden["segmem_v2_with_prev.gen32"])
    assert min(min(m) for m in m2 if m) > 1e-3, min(min(m) for m in m2 if m)     # measured 1.8e-3


def test_cached_greedy_decode_returns_the_reference_algorithms_tokens(golden):
    """oracle.t5_ref.generate_t5_cached (KV cache; the CPU baseline of the cached decode) == the recorded output of the
    reference's own no-cache `generate` on the golden inputs, token for token."""
    import torch
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    from oracle import t5_ref
    sd = {k: torch.from_numpy(v) for k, v in golden_weights(T5_SMALL).items()}
    mel = torch.from_numpy(synth_mel(2))
    with torch.no_grad():
        ids = t5_ref.generate_t5_cached(sd, T5_SMALL, mel, max_length=32)
    np.testing.assert_array_equal(ids.numpy(), golden["t5.gen32"])


def test_logmel_oracle_against_an_independent_float64_dft_and_analytic_triangles():
    """The frontend oracle is a restatement of torchaudio's algorithm through torch.stft + a transcription of
    melscale_fbanks; torchaudio is absent, so it stays "parity unpinned" — but a shared misuse of torch.stft (window
    periodicity, centring, normalisation, one-sided layout) or of the filterbank formula would go unnoticed by every
    test that compares the HIP kernel with it.  This check shares NOTHING with it: a direct O(N^2) DFT in float64 of the
    hann-windowed frames, and the HTK triangles evaluated analytically in float64 from the definition
    (contrib/spectrograms.py:128-145: n_fft 2048, hop 128, power 1, 512 mels from 20 to 7600 Hz, norm None)."""
    from oracle import logmel_ref
    rs = np.random.RandomState(11)
    audio = (rs.rand(2048 + 3 * 128).astype(np.float32) * 2 - 1)
    got = logmel_ref.compute_spectrogram(audio)                     # [frames, 512], natural log of the mel magnitudes
    n = np.arange(2048)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / 2048)                  # periodic hann
    k = np.arange(1025)
    dft = np.exp(-2j * np.pi * np.outer(k, n) / 2048)               # [1025, 2048]
    freqs = k * (16000 / 2) / 1024                                  # linspace(0, 8000, 1025)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    m_pts = np.linspace(mel(20.0), mel(7600.0), 514)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    fb = np.zeros((1025, 512))
    for m in range(512):
        lo, ce, hi = f_pts[m], f_pts[m + 1], f_pts[m + 2]
        fb[:, m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
    padded = np.concatenate([audio.astype(np.float64), np.zeros(4096)])
    for frame in (0, 1, 3):                                          # frame 3 still lies inside the real samples
        x = padded[frame * 128: frame * 128 + 2048] * win
        mag = np.abs(dft @ x)
        want = np.log(np.where(mag @ fb <= 0, 1e-5, mag @ fb))
        live = (mag @ fb) > 1e-3
        assert np.abs(got[frame][live] - want[live]).max() < 2e-4, (frame, np.abs(got[frame][live] - want[live]).max())
        assert (got[frame][~live] < np.log(2e-3)).all()
    # and the filterbank itself: same non-zero pattern and values as the restated torchaudio formula (f32)
    fb_ref = logmel_ref.melscale_fbanks().numpy()
    assert ((fb_ref > 0) == (fb > 1e-9)).mean() > 0.9999 and np.abs(fb_ref - fb).max() < 2e-4


def test_oracle_at_the_long_context_shape_matches_what_the_reference_recorded():
    """BASELINE configs[4] (config_slakh_segmem_finetune.yaml, mel_length 2048): the oracle's loss, sampled logits and EVERY
    gradient tensor (norm + 256 sampled elements) at B = 2 x 2048 frames (+64 memory slots) x 1024 tokens against what the
    reference itself produced at that shape (tests/golden/long_shape.npz, make_golden.py --long-shape) — the pin the GPU
    test of the same name (tests/test_bench_shape_gpu.py) stands on."""
    import os
    from mrmt3.synthetic import long_shape_inputs
    from conftest import GOLDEN
    fix = np.load(os.path.join(GOLDEN, "long_shape.npz"))
    variant = "segmem_v2_with_prev"
    torch.set_num_threads(8)
    mel, lab, prev = (torch.from_numpy(a) for a in long_shape_inputs())
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 1).items()}
    logits = t5_ref.forward_logits(sd, T5_SMALL, mel, lab, variant=variant, targets_prev=prev.clone())
    loss = t5_ref.ce_loss(logits, lab)
    loss.backward()
    idx = torch.from_numpy(fix[f"{variant}.logit_idx"])
    np.testing.assert_allclose(logits.detach().reshape(-1)[idx].numpy(), fix[f"{variant}.logit_val"], atol=5e-5, rtol=0)
    assert abs(loss.item() - float(fix[f"{variant}.fp32.loss"])) < 2e-5
    for n, norm, si, sv in zip(fix[f"{variant}.grad_names"].tolist(), fix[f"{variant}.grad_norm"],
                               fix[f"{variant}.grad_sample_idx"], fix[f"{variant}.grad_sample_val"]):
        g = sd[n].grad
        assert g is not None, n
        assert abs(g.double().norm().item() - norm) <= 1e-4 * norm + 1e-9, n
        np.testing.assert_allclose(g.reshape(-1)[torch.from_numpy(si)].numpy(), sv, rtol=0,
                                   atol=2e-4 * float(np.abs(sv).max()) + 1e-9, err_msg=n)
