"""`contrib.event_codec` — the MT3 event <-> index codec (reference contrib/event_codec.py:21-112).

Same public surface (`EventRange`, `Event`, `Codec.encode_event / decode_event_index /
event_type_range / is_shift_event_index / num_classes / max_shift_steps / steps_per_second`); the
lookup is table driven (prefix offsets + bisect) instead of a linear scan per call, because the
post-processing of a song decodes ~10^5 token ids.
"""
from __future__ import annotations

import bisect
import dataclasses
from typing import List, Tuple


@dataclasses.dataclass
class EventRange:
    type: str
    min_value: int
    max_value: int


@dataclasses.dataclass
class Event:
    type: str
    value: int


class Codec:
    """Index layout: the `shift` block first (starting at 0), then the given ranges in order."""

    def __init__(self, max_shift_steps: int, steps_per_second: float, event_ranges: List[EventRange]):
        self.steps_per_second = steps_per_second
        self._ranges = [EventRange("shift", 0, max_shift_steps)] + list(event_ranges)
        names = [r.type for r in self._ranges]
        if len(set(names)) != len(names):
            raise AssertionError("event types must be unique")
        self._starts, total = [], 0
        for r in self._ranges:
            self._starts.append(total)
            total += r.max_value - r.min_value + 1
        self._total = total
        self._by_name = {r.type: (s, r) for s, r in zip(self._starts, self._ranges)}

    @property
    def num_classes(self) -> int:
        return self._total

    @property
    def max_shift_steps(self) -> int:
        return self._ranges[0].max_value

    def is_shift_event_index(self, index: int) -> bool:
        return 0 <= index <= self._ranges[0].max_value

    def encode_event(self, event: Event) -> int:
        if event.type not in self._by_name:
            raise ValueError(f"Unknown event type: {event.type}")
        start, r = self._by_name[event.type]
        if not r.min_value <= event.value <= r.max_value:
            raise ValueError(f"Event value {event.value} is not within valid range "
                             f"[{r.min_value}, {r.max_value}] for type {event.type}")
        return start + event.value - r.min_value

    def event_type_range(self, event_type: str) -> Tuple[int, int]:
        if event_type not in self._by_name:
            raise ValueError(f"Unknown event type: {event_type}")
        start, r = self._by_name[event_type]
        return start, start + (r.max_value - r.min_value)

    def decode_event_index(self, index: int) -> Event:
        index = int(index)
        if not 0 <= index < self._total:
            raise ValueError(f"Unknown event index: {index}")
        i = bisect.bisect_right(self._starts, index) - 1
        r = self._ranges[i]
        return Event(type=r.type, value=r.min_value + index - self._starts[i])
