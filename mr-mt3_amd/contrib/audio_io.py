"""WAV reading for the evaluation driver — the two things test.py takes from `librosa.load(fname, sr=16000)`:
mono float32 samples in [-1, 1) and resampling to the model's rate.  PCM 8/16/24/32-bit and IEEE float WAV via
the standard library; resampling with a polyphase filter (scipy.signal.resample_poly).  librosa's default
resampler (soxr_hq) is a different filter, so resampled audio agrees with it only to filter accuracy — 16 kHz
input (the datasets are pre-resampled with the reference's resample.py) passes through untouched."""
from __future__ import annotations

import struct
import wave

import numpy as np


def read_wav(path: str):
    """-> (float32 samples [n] mono, sample rate)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt, data = None, None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                break
            tag, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            body = f.read(size + (size & 1))
            if tag == b"fmt ":
                fmt = body
            elif tag == b"data":
                data = body[:size]
        if fmt is None or data is None:
            raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == 0xFFFE and len(fmt) >= 26:                      # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first field
        code = struct.unpack("<H", fmt[24:26])[0]
    if code == 1:
        if bits == 8:
            x = (np.frombuffer(data, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data, "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(data, "<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif code == 3:
        x = np.frombuffer(data, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV encoding {code}")
    if channels > 1:
        x = x[: len(x) // channels * channels].reshape(-1, channels).mean(axis=1)      # librosa's mono=True
    return np.ascontiguousarray(x, dtype=np.float32), int(rate)


def load(path: str, sr: int = 16000):
    """`librosa.load(path, sr=sr)`: mono float32 at `sr`."""
    x, rate = read_wav(path)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(rate))
        x = resample_poly(x, sr // g, rate // g).astype(np.float32)
    return x, sr


def write_wav(path: str, samples, sr: int = 16000) -> None:
    """16-bit PCM mono (test fixtures and quick listening checks)."""
    pcm = (np.clip(np.asarray(samples, dtype=np.float64), -1.0, 32767.0 / 32768.0) * 32768.0).round().astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())
