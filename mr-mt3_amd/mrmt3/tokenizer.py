"""Target tokenisation for training rows (dataset/dataset_2_random.py:108-279).

notes of a recording -> one token stream indexed by audio frame (`tokenize`), then per training row:
cut the frames' tokens and prepend the "still sounding" state up to the tie token
(`extract_target_sequence`), collapse runs of single-step shifts into shift tokens whose value is the
time since the ROW START (not since the previous event — `dataset_2_random.py:237-244` re-emits the
running total) and drop repeated program / velocity changes (`run_length_encode_shifts`).  Host code
(numpy); the result feeds `mrmt3.batching.pad_targets`.  No note_seq: a recording's notes are a
`contrib.note_sequences.NoteSequence` (read one from a MIDI file with `contrib.midi_io`).
"""
from __future__ import annotations

import numpy as np

from contrib import event_codec, note_sequences, run_length_encoding, spectrograms, vocabularies


class Tokenizer:
    def __init__(self, codec=None, spectrogram_config=None, include_ties: bool = True, onsets_only: bool = False,
                 is_train: bool = True, is_randomize_tokens: bool = False, num_velocity_bins: int = 1):
        self.codec = codec or vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=num_velocity_bins))
        self.cfg = spectrogram_config or spectrograms.SpectrogramConfig()
        self.include_ties, self.onsets_only = include_ties, onsets_only
        self.is_train, self.is_randomize_tokens = is_train, is_randomize_tokens
        self.tie_token = self.codec.encode_event(event_codec.Event("tie", 0)) if include_ties else None
        self._state_ranges = [self.codec.event_type_range(t) for t in ("velocity", "program")]

    # ---- whole recording ---------------------------------------------------------------------------------
    def frame_times(self, n_samples: int) -> np.ndarray:
        """`_audio_to_frames` (:81-98): pad by hop - n % hop (a full hop when aligned), one time per frame."""
        hop = self.cfg.hop_width
        n_frames = (n_samples + hop - n_samples % hop) // hop
        return np.arange(n_frames) / self.cfg.frames_per_second

    def tokenize(self, ns: note_sequences.NoteSequence, n_samples: int) -> dict:
        """`_tokenize` (:108-172) after the tracks have been merged into `ns`."""
        note_sequences.assign_instruments(ns)
        note_sequences.validate_note_sequence(ns)
        if self.is_train:
            ns = note_sequences.trim_overlapping_notes(ns)
        if self.onsets_only:
            times, values = note_sequences.note_sequence_to_onsets(ns)
        else:
            times, values = note_sequences.note_sequence_to_onsets_and_offsets_and_programs(ns)
        ft = self.frame_times(n_samples)
        ev, start, end, st, st_idx = run_length_encoding.encode_and_index_events(
            state=note_sequences.NoteEncodingState() if self.include_ties else None,
            event_times=times, event_values=values, encode_event_fn=note_sequences.note_event_data_to_events,
            codec=self.codec, frame_times=ft,
            encoding_state_to_events_fn=note_sequences.note_encoding_state_to_events if self.include_ties else None)
        return {"input_times": ft, "targets": ev, "input_event_start_indices": start,
                "input_event_end_indices": end, "state_events": st, "input_state_event_indices": st_idx}

    # ---- one training row --------------------------------------------------------------------------------
    def extract_target_sequence(self, feats: dict, start_frame: int, n_frames: int) -> np.ndarray:
        """`_extract_target_sequence_with_indices` (:174-196) for frames [start_frame, start_frame+n_frames)."""
        lo = feats["input_event_start_indices"][start_frame]
        hi = feats["input_event_end_indices"][start_frame + n_frames - 1]
        targets = feats["targets"][lo:hi]
        if self.tie_token is not None:
            s = feats["input_state_event_indices"][start_frame]
            e = s + 1
            while feats["state_events"][e - 1] != self.tie_token:
                e += 1
            targets = np.concatenate([feats["state_events"][s:e], targets])
        return targets

    def _is_repeat(self, event: int, current: list) -> bool:
        """True when `event` sets a velocity / program that is already in force (updates `current`)."""
        for i, (lo, hi) in enumerate(self._state_ranges):
            if lo <= event <= hi:
                same = current[i] == event
                current[i] = event
                return same
        return False

    def run_length_encode_shifts(self, events) -> np.ndarray:
        """`_run_length_encode_shifts` (:198-250).  Trailing shifts are dropped with the row's end."""
        out, current = [], [0, 0]
        pending = total = 0
        for event in events:
            event = int(event)
            if self.codec.is_shift_event_index(event):
                pending += 1
                total += 1
                continue
            if not self.is_randomize_tokens and self._is_repeat(event, current):
                continue
            if pending > 0:
                left = total                              # time since the row start, in 10 ms steps
                while left > 0:
                    chunk = min(self.codec.max_shift_steps, left)
                    out.append(chunk)
                    left -= chunk
                pending = 0
            out.append(event)
        return np.asarray(out, dtype=np.int64)

    def remove_redundant_tokens(self, events) -> np.ndarray:
        """`_remove_redundant_tokens` (:252-279): the repeat filter alone (after token-order augmentation)."""
        current = [0, 0]
        return np.asarray([int(e) for e in events if not self._is_repeat(int(e), current)], dtype=np.int64)

    def randomize_tokens(self, events, rng=None) -> np.ndarray:
        """Token-order augmentation (`randomize_tokens`, :425-457): between two consecutive shift tokens the note
        groups — [program, velocity, pitch] or, for drums, [velocity, drum] — are put in a random order
        (`np.random.shuffle` of the group indices, same draw as the reference).  Everything before the first shift
        (the tie section) and from the last shift on is left alone.  Shift tokens are ids 0..999 as in the
        reference's `get_token_name` (id 1000, a full 10 s shift, is not treated as one there either)."""
        rng = rng or np.random
        ev = [int(e) for e in events]
        p_lo, p_hi = self.codec.event_type_range("program")
        v_lo, v_hi = self.codec.event_type_range("velocity")
        shifts = [i for i, e in enumerate(ev) if 0 <= e < 1000]
        if not shifts:
            return np.asarray(ev, dtype=np.int64)
        out = ev[:shifts[0]]
        for a, b in zip(shifts, shifts[1:]):
            out.append(ev[a])
            span, groups, k = ev[a + 1:b], [], 0
            while k < len(span):
                if p_lo <= span[k] <= p_hi:
                    n = 3
                elif v_lo <= span[k] <= v_hi:
                    n = 2
                else:
                    n = 1        # the reference would spin forever here; such spans do not occur after tokenisation
                groups.append(span[k:k + n])
                k += n
            order = np.arange(len(groups))
            rng.shuffle(order)
            for g in order:
                out.extend(groups[g])
        out.extend(ev[shifts[-1]:])
        return np.asarray(out, dtype=np.int64)

    def row_targets(self, feats: dict, start_frame: int, n_frames: int, rng=None) -> np.ndarray:
        """The target pipeline of `__getitem__` (:403-414) for one row: extract (+ tie section), run-length encode,
        and with `is_randomize_tokens` shuffle the note groups and only then drop repeated program / velocity tokens."""
        row = self.run_length_encode_shifts(self.extract_target_sequence(feats, start_frame, n_frames))
        if self.is_randomize_tokens:
            row = self.remove_redundant_tokens(self.randomize_tokens(row, rng))
        return row

    def targets_for_crop(self, feats: dict, rng=None):
        """`(start_frame, n_frames) -> token ids` for `mrmt3.batching.DeviceBatcher.build`."""
        return lambda start_frame, n_frames: self.row_targets(feats, start_frame, n_frames, rng)


def collate_fn(lst):
    """dataset_2_random.py:496-499: every item is already a [rows, ...] block; blocks are concatenated."""
    import torch
    return torch.cat([k[0] for k in lst]), torch.cat([k[1] for k in lst])
