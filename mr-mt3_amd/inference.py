"""`inference.InferenceHandler` — MI355X drop-in for the reference's inference driver up to token ids
(inference.py:20-215).  Same method names and return conventions; what changed is WHERE things run:

  * `_preprocess`: the padded audio goes to the GPU once and every 256-frame segment's log-mel,
    clip/scale and padded-frame zeroing is ONE `mrmt3_logmel_fwd` launch (the reference computes
    each segment's spectrogram on the CPU main process, rebuilding the filterbank per call).
  * `inference`: `model.generate` is the KV-cached hipGraph decoder; `_postprocess_batch` is
    unchanged arithmetic on the returned ids.
  * `_to_event` / MIDI writing (inference.py:217-234, 195-201): the codec, the run-length decoder and
    the note state machine are restated without note_seq/seqio (`contrib/{event_codec,vocabularies,
    run_length_encoding,note_sequences,metrics_utils,midi_io}.py`); `inference(..., outpath=...)`
    writes a Standard MIDI File and returns the note sequence.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from contrib import metrics_utils, midi_io, note_sequences, spectrograms, vocabularies

MIN_LOG_MEL = -12
MAX_LOG_MEL = 5
NUM_SPECIAL_TOKENS = 3     # PAD 0 / EOS 1 / UNK 2 (contrib/vocabularies.py:150-171)


def audio_to_frames(audio, spectrogram_config=None):
    """inference.py:64-75 — pads by `hop - len % hop` samples (a full hop when already aligned)."""
    cfg = spectrogram_config or spectrograms.SpectrogramConfig()
    frame_size = cfg.hop_width
    audio = np.pad(audio, [0, frame_size - len(audio) % frame_size], mode='constant')
    frames = spectrograms.split_audio(audio, cfg)
    num_frames = len(audio) // frame_size
    times = np.arange(num_frames) / cfg.frames_per_second
    return frames, times


def split_into_segments(frames, frame_times, max_length=256):
    """inference.py:77-95 — ceil(n/256) zero-padded segments and the count of real frames in each."""
    assert len(frames.shape) >= 1 and frames.shape[0] == frame_times.shape[0]
    num_segment = math.ceil(frames.shape[0] / max_length)
    batchs, times, paddings = [], [], []
    for i in range(num_segment):
        batch = np.zeros((max_length, *frames.shape[1:]))
        t = np.zeros((max_length))
        start = i * max_length
        end = max_length if start + max_length < frames.shape[0] else frames.shape[0] - start
        batch[0:end, ...] = frames[start:start + end, ...]
        t[0:end] = frame_times[start:start + end]
        batchs.append(batch), times.append(t), paddings.append(end)
    return np.stack(batchs, axis=0), np.stack(times, axis=0), paddings


def postprocess_batch(result: torch.Tensor, eos_token_id=1, num_special_tokens=NUM_SPECIAL_TOKENS):
    """inference.py:206-215 — positions at/after the first EOS -> -1, drop the 3 specials, drop BOS."""
    after_eos = torch.cumsum((result == eos_token_id).float(), dim=-1)
    result = result - num_special_tokens
    result = torch.where(after_eos.bool(), -1, result)
    return result[:, 1:].cpu().numpy()


class InferenceHandler:
    def __init__(self, model=None, weight_path=None, device=torch.device('cuda'), mel_norm=True,
                 contiguous_inference=False, use_tf_spectral_ops=False) -> None:
        if model is None:
            from models.t5 import T5ForConditionalGeneration
            from mrmt3.synthetic import T5_SMALL
            model = T5ForConditionalGeneration(T5_SMALL)
            model.load_state_dict(torch.load(weight_path, map_location='cpu'), strict=True)
            model.eval()
        if use_tf_spectral_ops:
            raise NotImplementedError("TF/ddsp spectral ops are out of scope (SURVEY §2.1 row 1)")
        self.model = model
        self.contiguous_inference = contiguous_inference
        self.SAMPLE_RATE = 16000
        self.spectrogram_config = spectrograms.SpectrogramConfig()
        self.codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))   # inference.py:52-53
        self.device = device
        self.model.to(self.device)
        self.mel_norm = mel_norm

    def _audio_to_frames(self, audio):
        return audio_to_frames(audio, self.spectrogram_config)

    def _split_token_into_length(self, frames, frame_times, max_length=256):
        return split_into_segments(frames, frame_times, max_length)

    def _compute_spectrograms(self, inputs, paddings=None):
        """[n_seg, 256, 128] frames -> ([n_seg, 256, 512] log-mel, raw samples); one kernel launch."""
        raw = np.reshape(inputs, (inputs.shape[0], -1))
        x = torch.from_numpy(raw).float().to(self.device)
        vf = None if paddings is None else torch.tensor(paddings, dtype=torch.int32, device=self.device)
        mel = spectrograms.logmel_segments(x, self.spectrogram_config, normalize=self.mel_norm, valid_frames=vf)
        return mel, raw

    def _preprocess(self, audio):
        frames, frame_times = self._audio_to_frames(audio)
        frames, frame_times, paddings = self._split_token_into_length(frames, frame_times)
        inputs, _ = self._compute_spectrograms(frames, paddings)     # padded frames zeroed in-kernel
        return inputs, frame_times

    def _batching(self, tensors, frame_times, batch_size=5):
        batchs, ft = [], []
        for start in range(0, tensors.shape[0], batch_size):
            end = min(start + batch_size, tensors.shape[0])
            batchs.append(tensors[start:end])
            ft.append(frame_times[start:end])
        return batchs, ft

    def _postprocess_batch(self, result):
        return postprocess_batch(result, self.model.config.eos_token_id)

    def _to_event(self, predictions_np, frame_times):
        """inference.py:217-234 — per segment: cut at the first decoded EOS (-1), segment start time =
        first frame time rounded down to the codec step, then decode all segments with ties."""
        predictions = []
        for i, batch in enumerate(predictions_np):
            for j, tokens in enumerate(batch):
                # NB (kept from the reference): argmax of an all-False mask is 0, so a segment that
                # never emitted EOS contributes NO tokens.
                tokens = tokens[:np.argmax(tokens == vocabularies.DECODED_EOS_ID)]
                start_time = frame_times[i][j][0]
                start_time -= start_time % (1 / self.codec.steps_per_second)
                predictions.append({"est_tokens": tokens, "start_time": start_time, "raw_inputs": []})
        result = metrics_utils.event_predictions_to_ns(predictions, codec=self.codec,
                                                       encoding_spec=note_sequences.NoteEncodingWithTiesSpec)
        return result["est_ns"]

    @torch.no_grad()
    def inference(self, audio, audio_path=None, outpath=None, valid_programs=None, num_beams=1, batch_size=5,
                  max_length=1024, verbose=False, return_tokens=False):
        """audio -> note sequence (and a MIDI file when `outpath` is given, like inference.py:149-204).
        `return_tokens=True` returns (post-processed token arrays per batch, frame times) instead."""
        inputs, frame_times = self._preprocess(audio)
        batches, ft = self._batching(inputs, frame_times, batch_size=batch_size)
        if self.contiguous_inference:
            batches = [torch.cat(batches, dim=0)]
            ft = [np.concatenate(ft, axis=0)]
        results = []
        for batch in batches:
            result = self.model.generate(inputs=batch.to(self.device), max_length=max_length)
            results.append(self._postprocess_batch(result))
        if return_tokens:
            return results, ft
        ns = self._to_event(results, ft)
        if outpath is not None:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(outpath)), exist_ok=True)
            midi_io.note_sequence_to_midi_file(ns, outpath)
        return ns

    @torch.no_grad()
    def inference_many(self, audios, outpaths=None, max_length=1024, return_tokens=False):
        """Several recordings in one go.  Segment-memory models decode them in lockstep (one batch row per
        recording, `model.generate_songs`); the plain T5 simply batches all segments.  Returns one note sequence
        (or, with `return_tokens`, one `(token arrays, frame times)` pair) per recording, like `inference`."""
        pre = [self._preprocess(a) for a in audios]
        if hasattr(self.model, "generate_songs"):
            ids = self.model.generate_songs([x.to(self.device) for x, _ in pre], max_length=max_length)
        else:
            flat = self.model.generate(inputs=torch.cat([x for x, _ in pre]).to(self.device), max_length=max_length)
            ids, at = [], 0
            for x, _ in pre:
                ids.append(flat[at:at + x.shape[0]])
                at += x.shape[0]
        out = []
        for k, (seg_ids, (_, ft)) in enumerate(zip(ids, pre)):
            results, times = [self._postprocess_batch(seg_ids)], [ft]
            if return_tokens:
                out.append((results, times))
                continue
            ns = self._to_event(results, times)
            if outpaths is not None and outpaths[k] is not None:
                import os
                os.makedirs(os.path.dirname(os.path.abspath(outpaths[k])), exist_ok=True)
                midi_io.note_sequence_to_midi_file(ns, outpaths[k])
            out.append(ns)
        return out
