"""`contrib.run_length_encoding` — `decode_events` (reference contrib/run_length_encoding.py:192-247):
shift tokens accumulate time (relative to the segment start, reset by any non-shift event), every
other token is handed to the state machine; and `encode_and_index_events` (`:81-189`), its inverse
for building training targets."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

from contrib import event_codec


def decode_events(state, tokens, start_time, max_time: Optional[float], codec, decode_event_fn: Callable
                  ) -> Tuple[int, int]:
    """Returns (invalid_events, dropped_events); `state` is updated in place."""
    invalid = dropped = 0
    steps = 0
    now = start_time
    n = len(tokens)
    for i in range(n):
        try:
            event = codec.decode_event_index(tokens[i])
        except ValueError:
            invalid += 1
            continue
        if event.type == "shift":
            steps += event.value
            now = start_time + steps / codec.steps_per_second
            if max_time and now > max_time:
                dropped = n - i
                break
        else:
            steps = 0
            try:
                decode_event_fn(state, now, event, codec)
            except ValueError:
                invalid += 1
    return invalid, dropped


def encode_and_index_events(state, event_times, event_values, encode_event_fn, codec, frame_times,
                            encoding_state_to_events_fn=None):
    """Timed events -> token stream with one `shift 1` per 10 ms step, indexed by audio frame
    (reference contrib/run_length_encoding.py:81-189).

    Returns (events, event_start_indices, event_end_indices, state_events, state_event_indices) as numpy
    arrays: frame f's targets are events[start[f]:end[f]] (end[f] == start[f+1]); state_events is the
    concatenation of "what is sounding" dumps taken before every event, state_event_indices[f] the dump
    valid at frame f.  Steps are `round(time * steps_per_second)` (Python rounding, half to even)."""
    import numpy as np
    order = np.argsort(np.asarray(event_times, dtype=np.float64), kind="stable")
    sps = codec.steps_per_second
    shift = codec.encode_event(event_codec.Event("shift", 1))
    n_frames = len(frame_times)
    events, state_events, starts, state_idx = [], [], [], []
    step = 0
    ev_mark = st_mark = 0            # len(events) / len(state_events) when the current step began

    def cover_frames():
        # frames that begin before the current step's time start at the marks of the step just finished
        while len(starts) < n_frames and frame_times[len(starts)] < step / sps:
            starts.append(ev_mark)
            state_idx.append(st_mark)

    for i in order:
        target = round(event_times[i] * sps)
        while step < target:
            events.append(shift)
            step += 1
            cover_frames()
            ev_mark, st_mark = len(events), len(state_events)
        if encoding_state_to_events_fn is not None:     # the state BEFORE this event
            state_events.extend(codec.encode_event(e) for e in encoding_state_to_events_fn(state))
        events.extend(codec.encode_event(e) for e in encode_event_fn(state, event_values[i], codec))
    # shifts up to and including the step that lands exactly on the last frame time
    while step / sps <= frame_times[-1]:
        events.append(shift)
        step += 1
        cover_frames()
        ev_mark = len(events)
    ends = starts[1:] + [len(events)]
    return (np.array(events), np.array(starts), np.array(ends), np.array(state_events), np.array(state_idx))
