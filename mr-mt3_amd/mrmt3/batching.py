"""GPU-resident batch construction from raw audio.

The reference builds every training row on a DataLoader worker CPU: split the song into chunks,
pick a run of chunks, pick a random mel_length window in each, compute that window's log-mel with
torchaudio, normalise, zero-pad, and ship 512 KB of mel per row (dataset/dataset_2_random.py:308-344,
281-306, 385-420).  Here the worker ships the recording ONCE (128 KB per 256 frames) and only the
crop DECISIONS are made on the host, with the same `random.randint` draws in the same order; the crops
are never materialised — `mrmt3_logmel_crops_fwd` reads each one straight out of the recording in
HBM and writes the padded `[B, mel_length, 512]` batch in one launch.

Target tokenisation (`_extract_target_sequence_with_indices`, `_run_length_encode_shifts`) stays on the
host and is passed in per row; `pad_targets` is the `_pad_length` target half.
"""
from __future__ import annotations

import random as _random
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class CropPlan:
    """One row per batch element, in frames (1 frame = hop_width samples) of the recording."""
    start_frame: np.ndarray      # int64 [B]
    valid_frames: np.ndarray     # int32 [B]  (< mel_length only for recordings shorter than one window)
    chunk_start: np.ndarray      # int64 [B]  start of the `_split_frame` chunk the crop was drawn from


def plan_crops(n_frames: int, mel_length: int = 256, num_rows_per_batch: int = 12, split_frame_length: int = 2000,
               is_deterministic: bool = False, rng=None) -> CropPlan:
    """Index arithmetic of `_split_frame` + row selection + `_random_chunk` for a recording of n_frames.
    `rng` is anything with `randint(a, b)` inclusive (default: the `random` module, like the reference), and
    is consumed in the reference's order: one draw for the run of chunks, then one per row."""
    rng = rng or _random
    chunks = [(s, split_frame_length) for s in range(0, n_frames, split_frame_length)
              if not s + split_frame_length >= n_frames]                    # last chunk dropped (:315-316)
    if not chunks:
        chunks = [(0, n_frames)]                                            # short song: the whole row (:325-326)
    if len(chunks) > num_rows_per_batch:
        first = 0 if is_deterministic else rng.randint(0, len(chunks) - num_rows_per_batch)
        chunks = chunks[first:first + num_rows_per_batch]
    starts, valid, cstart = [], [], []
    for s, n in chunks:
        slack = n - mel_length
        off = 0
        if slack >= 1 and not is_deterministic:                             # :332-338
            off = rng.randint(0, slack)
        starts.append(s + off)
        valid.append(min(mel_length, n))                                    # `_pad_length` zero-pads the rest
        cstart.append(s)
    return CropPlan(np.asarray(starts, np.int64), np.asarray(valid, np.int32), np.asarray(cstart, np.int64))


def pad_targets(targets, event_length: int = 1024, num_special_tokens: int = 3) -> torch.Tensor:
    """`_pad_length` target half (:294-305): truncate, +3, EOS, -100 padding; a target that already fills
    event_length gets neither EOS nor padding.  `targets`: list of 1-D int arrays -> int64 [B, event_length]."""
    out = np.full((len(targets), event_length), -100, np.int64)
    for i, t in enumerate(targets):
        t = np.asarray(t[:event_length], np.int64) + num_special_tokens
        out[i, :len(t)] = t
        if len(t) < event_length:
            out[i, len(t)] = 1
    return torch.from_numpy(out)


class DeviceBatcher:
    """recording on the GPU + crop plan -> `(inputs [B, mel_length, 512], targets [B, event_length])`, the
    pair `__getitem__` returns (:420), with `inputs` produced by one kernel launch."""

    def __init__(self, device, mel_length: int = 256, event_length: int = 1024, num_rows_per_batch: int = 12,
                 split_frame_length: int = 2000, is_deterministic: bool = False, spectrogram_config=None,
                 out_bf16: bool = False, rng=None):
        from contrib import spectrograms
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatcher builds batches on the GPU (no CPU fallback)")
        self.cfg = spectrogram_config or spectrograms.SpectrogramConfig()
        self.mel_length, self.event_length = mel_length, event_length
        self.num_rows, self.split_len = num_rows_per_batch, split_frame_length
        self.is_deterministic, self.out_bf16, self.rng = is_deterministic, out_bf16, rng
        self._sp = spectrograms

    def upload(self, samples) -> torch.Tensor:
        """host recording (numpy / tensor, any float dtype) -> f32 device tensor, padded to whole frames like
        `split_audio` (contrib/spectrograms.py:79-90).  Pinned staging + async copy."""
        x = torch.as_tensor(np.asarray(samples), dtype=torch.float32).reshape(-1)
        hop = self.cfg.hop_width
        pad = (-x.numel()) % hop
        if pad:
            x = torch.cat([x, x.new_zeros(pad)])
        return x.pin_memory().to(self.device, non_blocking=True)

    def plan(self, n_frames: int) -> CropPlan:
        return plan_crops(n_frames, self.mel_length, self.num_rows, self.split_len, self.is_deterministic, self.rng)

    def mel(self, audio_dev: torch.Tensor, plan: CropPlan) -> torch.Tensor:
        starts = torch.from_numpy(plan.start_frame)
        vf = torch.from_numpy(plan.valid_frames)
        return self._sp.logmel_crops(audio_dev, starts, self.mel_length, self.cfg, normalize=True,
                                     valid_frames=vf, out_bf16=self.out_bf16)

    def build(self, audio_dev: torch.Tensor, targets_for_crop, plan: CropPlan | None = None):
        """`targets_for_crop(start_frame, n_frames) -> 1-D int array` is the host tokeniser for one crop."""
        plan = plan or self.plan(audio_dev.numel() // self.cfg.hop_width)
        mel = self.mel(audio_dev, plan)
        tg = pad_targets([targets_for_crop(int(s), int(v)) for s, v in zip(plan.start_frame, plan.valid_frames)],
                         self.event_length)
        return mel, tg.to(self.device, non_blocking=True)
