"""ORACLE (test infrastructure, not product code) — CPU restatement of the log-mel frontend and of
the inference harness either side of it.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Follows, line by line:
  * `contrib/spectrograms.py:35-41`   constants (F1)
  * `contrib/spectrograms.py:92-98`   pad_end (F3)
  * `contrib/spectrograms.py:100-103` safe_log (F5)
  * `contrib/spectrograms.py:128-145` compute_spectrogram, torch branch (F4)
  * `dataset/dataset_2_random.py:288-289` == `inference.py:115-117` clip + scale (F6)
  * `inference.py:64-136,206-215`     InferenceHandler pre/post-processing (I1-I3, F2, F7)

The mel arithmetic itself lives in a THIRD-PARTY dependency that is absent from /root/reference
and from this image: `torchaudio.transforms.MelSpectrogram` (README.md:25 lists "torchaudio",
version unpinned).  Its published algorithm is restated here:
  Spectrogram  = torch.stft(x, n_fft=2048, hop_length=128, win_length=2048,
                            window=hann_window(2048, periodic=True), center=False,
                            normalized=False, onesided=True).abs()          (power=1.0)
  MelScale     = spec^T @ melscale_fbanks(n_freqs=1025, f_min=20, f_max=7600, n_mels=512,
                            sample_rate=16000, norm=None, mel_scale="htk")
PARITY PINNING: the reference has no test or golden vector for the frontend and torchaudio cannot
be imported here, so the frontend oracle is pinned only by known-answer anchors derived from the
cited lines (see tests/test_oracle_golden.py): output shape, pad 1920, exactly two all-zero
filters, 1934 filterbank non-zeros, constant log(1e-5) columns, pure-tone peak bin.  Treat it as
"parity unpinned (third-party algorithm restated)".
"""
from __future__ import annotations

import math

import numpy as np
import torch

SAMPLE_RATE = 16000
HOP_WIDTH = 128
NUM_MEL_BINS = 512
FFT_SIZE = 2048
MEL_LO_HZ = 20.0
MEL_HI_HZ = 7600.0
MIN_LOG_MEL = -12
MAX_LOG_MEL = 5


def hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs=FFT_SIZE // 2 + 1, f_min=MEL_LO_HZ, f_max=MEL_HI_HZ,
                    n_mels=NUM_MEL_BINS, sample_rate=SAMPLE_RATE) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'), fp32 torch ops."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))  # [n_freqs, n_mels]


def pad_end(samples: torch.Tensor, n_fft=FFT_SIZE, hop=HOP_WIDTH) -> torch.Tensor:
    n = samples.shape[-1]
    n_frames = -(-n // hop)
    pad = max(0, n_fft + hop * (n_frames - 1) - n)
    return torch.nn.functional.pad(samples, (0, pad))


def safe_log(x: torch.Tensor, eps=1e-5) -> torch.Tensor:
    return torch.log(torch.where(x <= 0.0, torch.full_like(x, eps), x))


def compute_spectrogram(samples: np.ndarray, fb: torch.Tensor | None = None) -> np.ndarray:
    """[N] audio -> [ceil(N/128), 512] log-mel (un-normalised), as `compute_spectrogram(...)`
    returns `S.numpy().T`.  `fb=None` rebuilds the filterbank per call like the reference does."""
    if fb is None:
        fb = melscale_fbanks()
    x = pad_end(torch.from_numpy(np.asarray(samples)).float())
    spec = torch.stft(x, FFT_SIZE, hop_length=HOP_WIDTH, win_length=FFT_SIZE,
                      window=torch.hann_window(FFT_SIZE), center=False, normalized=False,
                      onesided=True, return_complex=True).abs()          # [1025, frames]
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)     # [512, frames]
    return safe_log(mel).numpy().T


def normalize_mel(mel: np.ndarray) -> np.ndarray:
    mel = np.clip(mel, MIN_LOG_MEL, MAX_LOG_MEL)
    return (mel - MIN_LOG_MEL) / (MAX_LOG_MEL - MIN_LOG_MEL)


def logmel_segments(audio: np.ndarray, fb: torch.Tensor | None = None) -> np.ndarray:
    """[B, n] -> [B, n/128, 512] normalised log-mel; each segment is padded on its own (the last
    15 of 256 frames see zeros, not the next segment)."""
    if fb is None:
        fb = melscale_fbanks()
    return np.stack([normalize_mel(compute_spectrogram(a, fb)) for a in audio]).astype(np.float32)


# --- inference harness (inference.py) ------------------------------------------------------------

def audio_to_frames(audio: np.ndarray):
    """`inference.py:64-75`: pads by hop - len%hop (a FULL hop when already aligned)."""
    pad = HOP_WIDTH - len(audio) % HOP_WIDTH
    audio = np.pad(audio, [0, pad], mode="constant")
    frames = audio.reshape(-1, HOP_WIDTH)
    times = np.arange(len(audio) // HOP_WIDTH) / (SAMPLE_RATE / HOP_WIDTH)
    return frames, times


def split_into_segments(frames, frame_times, max_length=256):
    """`inference.py:77-95`."""
    n = frames.shape[0]
    n_seg = math.ceil(n / max_length)
    batches, times, paddings = [], [], []
    for i in range(n_seg):
        b = np.zeros((max_length, *frames.shape[1:]))
        t = np.zeros((max_length))
        start = i * max_length
        end = max_length if start + max_length < n else n - start
        b[0:end] = frames[start:start + end]
        t[0:end] = frame_times[start:start + end]
        batches.append(b), times.append(t), paddings.append(end)
    return np.stack(batches), np.stack(times), paddings


def preprocess(audio: np.ndarray, mel_norm=True):
    """`inference.py:120-127`."""
    frames, ft = audio_to_frames(audio)
    frames, ft, paddings = split_into_segments(frames, ft)
    fb = melscale_fbanks()
    mel = np.stack([compute_spectrogram(f.reshape(-1), fb) for f in frames])
    if mel_norm:
        mel = normalize_mel(mel)
    for i, p in enumerate(paddings):
        mel[i, p:] = 0
    return mel, ft, paddings


def postprocess_batch(result: np.ndarray, eos_id=1, num_special=3) -> np.ndarray:
    """`inference.py:206-215`."""
    after_eos = np.cumsum((result == eos_id).astype(np.float32), axis=-1)
    out = result - num_special
    out = np.where(after_eos.astype(bool), -1, out)
    return out[:, 1:]


# ---- producer side of the training dataset (dataset/dataset_2_random.py) -----------------------------------
# The dataset module itself cannot be imported here (librosa / note_seq / seqio absent): these are
# line-by-line restatements, PARITY UNPINNED by reference outputs, pinned by known-answer tests only.
def split_frame(row: dict, length=2000) -> list:
    """`_split_frame` (dataset_2_random.py:308-327): chunks of `length` frames; the last (partial OR
    exactly full) chunk is dropped (`split + length >= input_length`); a song shorter than one chunk
    is returned whole."""
    per_frame = ("inputs", "input_times", "input_event_start_indices", "input_event_end_indices",
                 "input_state_event_indices")
    rows = []
    n = row["inputs"].shape[0]
    for split in range(0, n, length):
        if split + length >= n:
            continue
        rows.append({k: (v[split:split + length] if k in per_frame else v) for k, v in row.items()})
    return rows if rows else [row]


def select_rows(rows: list, num_rows_per_batch: int, rng, is_deterministic=False) -> list:
    """`__getitem__` (`:395-400`): a random run of `num_rows_per_batch` consecutive chunks."""
    if len(rows) > num_rows_per_batch:
        start = 0 if is_deterministic else rng.randint(0, len(rows) - num_rows_per_batch)
        rows = rows[start:start + num_rows_per_batch]
    return rows


def random_chunk(row: dict, mel_length: int, rng, is_deterministic=False) -> dict:
    """`_random_chunk` (`:329-344`): a random window of mel_length frames inside the chunk
    (`random.randint` is inclusive on both ends)."""
    per_frame = ("inputs", "input_times", "input_event_start_indices", "input_event_end_indices",
                 "input_state_event_indices")
    n = row["inputs"].shape[0]
    random_length = n - mel_length
    if random_length < 1:
        return row
    start = 0 if is_deterministic else rng.randint(0, random_length)
    return {k: (v[start:start + mel_length] if k in per_frame else v) for k, v in row.items()}


def compute_spectrogram_row(frames: np.ndarray, fb=None) -> np.ndarray:
    """`_compute_spectrogram` (`:281-290`): flatten the row's frames, log-mel, clip to [-12, 5], scale."""
    return normalize_mel(compute_spectrogram(np.asarray(frames, dtype=np.float32).reshape(-1), fb))


def pad_length(mel: np.ndarray, targets: np.ndarray, mel_length: int, event_length: int, num_special=3):
    """`_pad_length` (`:292-306`): mel truncated / zero-padded to mel_length rows; targets truncated to
    event_length, shifted by the 3 special ids, then EOS(1) and -100 padding IF shorter than
    event_length (a full-length target gets no EOS)."""
    inputs = np.asarray(mel[:mel_length], dtype=np.float32)
    t = np.asarray(targets[:event_length], dtype=np.int64) + num_special
    if inputs.shape[0] < mel_length:
        inputs = np.concatenate([inputs, np.zeros((mel_length - inputs.shape[0], inputs.shape[1]), np.float32)], 0)
    if t.shape[0] < event_length:
        n_pad = event_length - t.shape[0] - 1
        t = np.concatenate([t, [1]] + ([np.full(n_pad, -100, np.int64)] if n_pad > 0 else []))
    return inputs, t
