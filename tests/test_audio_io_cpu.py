"""contrib.audio_io — the WAV reading the evaluation driver uses instead of librosa.load."""
import struct

import numpy as np
import pytest

from contrib import audio_io


def _riff(fmt_body, data):
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
             b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_pcm16_round_trip_and_extra_chunks(tmp_path):
    x = np.sin(2 * np.pi * 440 * np.arange(16000) / 16000) * 0.5
    p = str(tmp_path / "a.wav")
    audio_io.write_wav(p, x)
    y, sr = audio_io.load(p)
    assert sr == 16000 and y.dtype == np.float32 and len(y) == 16000
    assert np.abs(y - x).max() < 1.0 / 32768
    # a LIST chunk between fmt and data, stereo 24-bit: channels averaged, sign handled
    l = np.array([0.5, -0.5, 0.25], np.float64)
    r = np.array([0.5, 0.5, -0.75], np.float64)
    pcm = b""
    for a, b in zip(l, r):
        for v in (a, b):
            pcm += int(round(v * 8388608)).to_bytes(3, "little", signed=True)
    fmt = struct.pack("<HHIIHH", 1, 2, 22050, 22050 * 6, 6, 24)
    (tmp_path / "b.wav").write_bytes(_riff(fmt, pcm))
    z, rate = audio_io.read_wav(str(tmp_path / "b.wav"))
    assert rate == 22050 and np.allclose(z, (l + r) / 2, atol=1e-6)
    # IEEE float
    f = np.array([0.1, -0.2, 0.3], "<f4")
    (tmp_path / "c.wav").write_bytes(_riff(struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32), f.tobytes()))
    assert np.array_equal(audio_io.read_wav(str(tmp_path / "c.wav"))[0], f)
    (tmp_path / "d.wav").write_bytes(b"RIFX0000WAVE")
    with pytest.raises(ValueError):
        audio_io.read_wav(str(tmp_path / "d.wav"))


def test_resampling_keeps_duration_and_pitch(tmp_path):
    sr0 = 44100
    t = np.arange(sr0 * 2) / sr0
    x = 0.4 * np.sin(2 * np.pi * 1000.0 * t)
    p = str(tmp_path / "hi.wav")
    audio_io.write_wav(p, x, sr0)
    y, sr = audio_io.load(p, sr=16000)
    assert sr == 16000 and abs(len(y) - 32000) <= 1
    spec = np.abs(np.fft.rfft(y[:16000] * np.hanning(16000)))
    assert abs(int(spec.argmax()) - 1000) <= 1 and abs(np.abs(y[2000:30000]).max() - 0.4) < 0.01
