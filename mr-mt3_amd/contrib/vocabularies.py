"""`contrib.vocabularies` — the slice of the reference module the transcription path needs after the
model: vocabulary constants and the event codec (reference contrib/vocabularies.py:28-34,54-68,118-139,
150-171).  The seqio/t5 `Vocabulary` classes are not needed on this side: ids leave the decoder
already shifted by `num_special_tokens` in `inference.postprocess_batch`.
"""
from __future__ import annotations

import dataclasses
import math

from contrib import event_codec

DECODED_EOS_ID = -1
DECODED_INVALID_ID = -2

DEFAULT_STEPS_PER_SECOND = 100
DEFAULT_MAX_SHIFT_SECONDS = 10
DEFAULT_NUM_VELOCITY_BINS = 127

MIN_MIDI_PITCH, MAX_MIDI_PITCH = 0, 127          # note_seq constants
MIN_MIDI_PROGRAM, MAX_MIDI_PROGRAM = 0, 127
MAX_MIDI_VELOCITY = 127
NUM_SPECIAL_TOKENS = 3                           # PAD 0, EOS 1, UNK 2
DEFAULT_EXTRA_IDS = 100                          # t5.data.DEFAULT_EXTRA_IDS


@dataclasses.dataclass
class VocabularyConfig:
    """Vocabulary configuration parameters."""
    steps_per_second: int = DEFAULT_STEPS_PER_SECOND
    max_shift_seconds: int = DEFAULT_MAX_SHIFT_SECONDS
    num_velocity_bins: int = DEFAULT_NUM_VELOCITY_BINS


def num_velocity_bins_from_codec(codec: event_codec.Codec) -> int:
    lo, hi = codec.event_type_range("velocity")
    return hi - lo


def velocity_to_bin(velocity: int, num_velocity_bins: int) -> int:
    return 0 if velocity == 0 else math.ceil(num_velocity_bins * velocity / MAX_MIDI_VELOCITY)


def bin_to_velocity(velocity_bin: int, num_velocity_bins: int) -> int:
    return 0 if velocity_bin == 0 else int(MAX_MIDI_VELOCITY * velocity_bin / num_velocity_bins)


def build_codec(vocab_config: VocabularyConfig) -> event_codec.Codec:
    """shift [0, 100*10] | pitch 0-127 | velocity 0..bins (0 = note-off) | tie | program 0-127 | drum 0-127.
    With `num_velocity_bins=1` (inference.py:52-53) that is 1388 classes; +3 specials +100 extra ids,
    padded to the model's 1536-way head."""
    ranges = [
        event_codec.EventRange("pitch", MIN_MIDI_PITCH, MAX_MIDI_PITCH),
        event_codec.EventRange("velocity", 0, vocab_config.num_velocity_bins),
        event_codec.EventRange("tie", 0, 0),
        event_codec.EventRange("program", MIN_MIDI_PROGRAM, MAX_MIDI_PROGRAM),
        event_codec.EventRange("drum", MIN_MIDI_PITCH, MAX_MIDI_PITCH),
    ]
    return event_codec.Codec(max_shift_steps=vocab_config.steps_per_second * vocab_config.max_shift_seconds,
                             steps_per_second=vocab_config.steps_per_second, event_ranges=ranges)


def vocab_size(codec: event_codec.Codec, extra_ids: int = DEFAULT_EXTRA_IDS) -> int:
    """`GenericTokenVocabulary(codec.num_classes, extra_ids).vocab_size`."""
    return NUM_SPECIAL_TOKENS + codec.num_classes + extra_ids
