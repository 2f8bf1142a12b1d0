"""`contrib.run_length_encoding.decode_events` — the decode half of the reference module
(contrib/run_length_encoding.py:192-247): shift tokens accumulate time (relative to the segment
start, reset by any non-shift event), every other token is handed to the state machine."""
from __future__ import annotations

from typing import Callable, Optional, Tuple


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
