"""Audio spectrogram functions — MI355X drop-in for the reference's `contrib/spectrograms.py`.

Same names, arguments and return conventions as the reference module (`SpectrogramConfig`
:44-65, `split_audio` :68-90, `compute_spectrogram` :105-145, `flatten_frames` :148-155,
`input_depth` :158-159), but the log-mel arithmetic runs in the gfx950 kernel `mrmt3_logmel_fwd`
(csrc/logmel.hip) instead of `torchaudio.transforms.MelSpectrogram` on a CPU DataLoader worker.
There is no CPU fallback: without the HIP library / a GPU these functions raise.

Only the reference's PyTorch branch (`use_tf_spectral_ops=False`) is provided; the TF/ddsp branch
exists in the reference solely for the official `mt3.pth` checkpoint.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
import torch

from mrmt3 import lib

# defaults for spectrogram config (reference :35-41)
DEFAULT_SAMPLE_RATE = 16000
DEFAULT_HOP_WIDTH = 128
DEFAULT_NUM_MEL_BINS = 512
FFT_SIZE = 2048
MEL_LO_HZ = 20.0
MEL_HI_HZ = 7600.0


@dataclasses.dataclass
class SpectrogramConfig:
    """Spectrogram configuration parameters."""
    sample_rate: int = DEFAULT_SAMPLE_RATE
    hop_width: int = DEFAULT_HOP_WIDTH
    num_mel_bins: int = DEFAULT_NUM_MEL_BINS
    use_tf_spectral_ops: bool = False

    @property
    def abbrev_str(self):
        s = ''
        if self.sample_rate != DEFAULT_SAMPLE_RATE:
            s += 'sr%d' % self.sample_rate
        if self.hop_width != DEFAULT_HOP_WIDTH:
            s += 'hw%d' % self.hop_width
        if self.num_mel_bins != DEFAULT_NUM_MEL_BINS:
            s += 'mb%d' % self.num_mel_bins
        return s

    @property
    def frames_per_second(self):
        return self.sample_rate / self.hop_width


def split_audio(samples, spectrogram_config):
    """Split audio into hop-sized frames: [N] -> [ceil(N/hop), hop] (zero padded tail)."""
    hop = spectrogram_config.hop_width
    samples = np.asarray(samples)
    if samples.shape[0] % hop != 0:
        samples = np.pad(samples, (0, hop - samples.shape[0] % hop), 'constant', constant_values=0)
    return samples.reshape(-1, hop)


def flatten_frames(frames, use_tf_spectral_ops=False):
    """Convert frames back into a flat array of samples."""
    return np.reshape(frames, (-1,))


def input_depth(spectrogram_config):
    return spectrogram_config.num_mel_bins


# ---- kernel tables ---------------------------------------------------------------------------------

def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """HTK triangular filters, norm=None — the matrix torchaudio's MelScale multiplies by, built
    with the same fp32 torch ops; returns [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


_TABLES = {}


def kernel_tables(cfg: SpectrogramConfig, device):
    """Window, FFT twiddles and the filterbank in compressed-row form, resident on `device`.
    Built once per (config, device); the reference rebuilds its filterbank on every call."""
    key = (cfg.sample_rate, cfg.hop_width, cfg.num_mel_bins, str(device))
    if key in _TABLES:
        return _TABLES[key]
    fb = mel_filterbank(FFT_SIZE // 2 + 1, MEL_LO_HZ, MEL_HI_HZ, cfg.num_mel_bins, cfg.sample_rate).numpy()
    nz = fb > 0
    start = np.zeros(cfg.num_mel_bins, np.int32)
    cnt = np.zeros(cfg.num_mel_bins, np.int32)
    for m in range(cfg.num_mel_bins):
        idx = np.nonzero(nz[:, m])[0]
        if len(idx):
            start[m], cnt[m] = idx[0], idx[-1] - idx[0] + 1
    max_taps = max(int(cnt.max()), 1)
    w = np.zeros((cfg.num_mel_bins, max_taps), np.float32)
    for m in range(cfg.num_mel_bins):
        w[m, :cnt[m]] = fb[start[m]:start[m] + cnt[m], m]
    k = np.arange(FFT_SIZE // 2, dtype=np.float64)
    tw = np.stack([np.cos(-2 * np.pi * k / FFT_SIZE), np.sin(-2 * np.pi * k / FFT_SIZE)], axis=1).astype(np.float32)
    t = dict(
        hop=cfg.hop_width, n_mels=cfg.num_mel_bins, max_taps=max_taps, nnz=int(nz.sum()),
        window=torch.hann_window(FFT_SIZE).to(device),
        twiddle=torch.from_numpy(tw).to(device),
        fb_start=torch.from_numpy(start).to(device), fb_cnt=torch.from_numpy(cnt).to(device),
        fb_w=torch.from_numpy(w).to(device),
    )
    _TABLES[key] = t
    return t


def logmel_segments(audio: torch.Tensor, cfg: SpectrogramConfig = SpectrogramConfig(), normalize=True,
                    valid_frames=None, out_bf16=False) -> torch.Tensor:
    """Batched device entry point: audio [B, n] f32 on the GPU -> [B, ceil(n/hop), n_mels].
    Each row is padded on its own like `pad_end` does (frames near the end see zeros)."""
    if not audio.is_cuda:
        raise RuntimeError("logmel_segments needs a device tensor (no CPU fallback)")
    audio = audio.contiguous().float()
    return lib.logmel(audio, kernel_tables(cfg, audio.device), valid_frames=valid_frames, normalize=normalize,
                      out_bf16=out_bf16)


def logmel_crops(audio: torch.Tensor, start_frames: torch.Tensor, n_frames: int,
                 cfg: SpectrogramConfig = SpectrogramConfig(), normalize=True, valid_frames=None,
                 out_bf16=False) -> torch.Tensor:
    """Crops of ONE recording resident on the GPU -> [B, n_frames, n_mels], without materialising the
    crops: `start_frames` [B] (int64, in hops) are the `_random_chunk` offsets
    (dataset/dataset_2_random.py:329-344); each crop is transformed on its own like
    `_compute_spectrogram` (`:281-290`) does, and frames >= valid_frames[b] come out zero like the
    rows `_pad_length` appends (`:295-297`)."""
    if not audio.is_cuda:
        raise RuntimeError("logmel_crops needs a device tensor (no CPU fallback)")
    starts = (start_frames.to(device=audio.device, dtype=torch.int64) * cfg.hop_width).contiguous()
    vf = None if valid_frames is None else valid_frames.to(device=audio.device, dtype=torch.int32).contiguous()
    return lib.logmel_crops(audio.contiguous().float().view(-1), starts, n_frames * cfg.hop_width,
                            kernel_tables(cfg, audio.device), valid_frames=vf, normalize=normalize,
                            out_bf16=out_bf16)


def compute_spectrogram(samples, spectrogram_config, device=None):
    """[N] samples (numpy) -> [ceil(N/hop), n_mels] log-mel (numpy, un-normalised), like the
    reference function; runs on `device` (default cuda:0)."""
    if spectrogram_config.use_tf_spectral_ops:
        raise NotImplementedError("TF/ddsp spectral ops are out of scope (SURVEY §2.1 row 1)")
    dev = torch.device(device or "cuda:0")
    x = torch.from_numpy(np.asarray(samples)).float().to(dev)[None]
    return logmel_segments(x, spectrogram_config, normalize=False)[0].cpu().numpy()
