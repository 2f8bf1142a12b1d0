"""`contrib.metrics_utils.event_predictions_to_ns` — combine per-segment token predictions into one
note sequence (reference contrib/metrics_utils.py:50-143)."""
from __future__ import annotations

import functools

import numpy as np

from contrib import run_length_encoding


def decode_and_combine_predictions(predictions, init_state_fn, begin_segment_fn, decode_tokens_fn, flush_state_fn):
    """Segments sorted by start time; a segment may not emit events at/after the next segment's start."""
    ordered = sorted(predictions, key=lambda p: p["start_time"])
    state = init_state_fn()
    invalid = dropped = 0
    for i, pred in enumerate(ordered):
        begin_segment_fn(state)
        limit = ordered[i + 1]["start_time"] if i + 1 < len(ordered) else None
        a, b = decode_tokens_fn(state, pred["est_tokens"], pred["start_time"], limit)
        invalid += a
        dropped += b
    return flush_state_fn(state), invalid, dropped


def event_predictions_to_ns(predictions, codec, encoding_spec):
    ns, invalid, dropped = decode_and_combine_predictions(
        predictions, encoding_spec.init_decoding_state_fn, encoding_spec.begin_decoding_segment_fn,
        functools.partial(run_length_encoding.decode_events, codec=codec, decode_event_fn=encoding_spec.decode_event_fn),
        encoding_spec.flush_decoding_state_fn)
    ordered = sorted(predictions, key=lambda p: p["start_time"])
    raws = [np.asarray(p["raw_inputs"]) for p in ordered]
    return {"raw_inputs": np.concatenate(raws, axis=0) if raws else np.zeros(0),
            "start_times": [p["start_time"] for p in ordered], "est_ns": ns,
            "est_invalid_events": invalid, "est_dropped_events": dropped}
