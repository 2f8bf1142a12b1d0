"""`contrib.note_sequences` — token events -> notes (reference contrib/note_sequences.py:24-28,68-80,
258-408), without the note_seq protobuf: a `NoteSequence` here is a plain list of `Note` records plus
`total_time`, which is all the decoding state machine and the MIDI writer use.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Set, Tuple

from contrib import event_codec, vocabularies

DEFAULT_VELOCITY = 100
DEFAULT_NOTE_DURATION = 0.01
MIN_NOTE_DURATION = 0.01


@dataclasses.dataclass
class Note:
    start_time: float
    end_time: float
    pitch: int
    velocity: int
    program: int = 0
    is_drum: bool = False
    instrument: int = 0


@dataclasses.dataclass
class NoteSequence:
    notes: List[Note] = dataclasses.field(default_factory=list)
    total_time: float = 0.0
    ticks_per_quarter: int = 220


@dataclasses.dataclass
class NoteDecodingState:
    """Decoding state for note transcription (reference :258-277)."""
    current_time: float = 0.0
    current_velocity: int = DEFAULT_VELOCITY          # 0 = the following pitches are note-offs
    current_program: int = 0
    active_pitches: Dict[Tuple[int, int], Tuple[float, int]] = dataclasses.field(default_factory=dict)
    tied_pitches: Set[Tuple[int, int]] = dataclasses.field(default_factory=set)
    is_tie_section: bool = False
    note_sequence: NoteSequence = dataclasses.field(default_factory=NoteSequence)


def _emit(ns: NoteSequence, start, end, pitch, velocity, program=0, is_drum=False):
    end = max(end, start + MIN_NOTE_DURATION)
    ns.notes.append(Note(start, end, int(pitch), int(velocity), int(program), is_drum))
    ns.total_time = max(ns.total_time, end)


def decode_note_event(state: NoteDecodingState, time: float, event: event_codec.Event,
                      codec: event_codec.Codec) -> None:
    """One non-shift event (reference :305-372).  Raises ValueError for events that are invalid in
    the current state; the caller counts those and moves on."""
    if time < state.current_time:
        raise ValueError("event time < current time, %f < %f" % (time, state.current_time))
    state.current_time = time
    kind = event.type
    if kind == "pitch":
        key = (event.value, state.current_program)
        if state.is_tie_section:
            if key not in state.active_pitches:
                raise ValueError("inactive pitch/program in tie section: %d/%d" % key)
            if key in state.tied_pitches:
                raise ValueError("pitch/program is already tied: %d/%d" % key)
            state.tied_pitches.add(key)
        elif state.current_velocity == 0:
            if key not in state.active_pitches:
                raise ValueError("note-off for inactive pitch/program: %d/%d" % key)
            onset, vel = state.active_pitches.pop(key)
            _emit(state.note_sequence, onset, time, key[0], vel, key[1])
        else:
            if key in state.active_pitches:       # re-onset: close the running note first
                onset, vel = state.active_pitches.pop(key)
                _emit(state.note_sequence, onset, time, key[0], vel, key[1])
            state.active_pitches[key] = (time, state.current_velocity)
    elif kind == "drum":
        if state.current_velocity == 0:
            raise ValueError("velocity cannot be zero for drum event")
        _emit(state.note_sequence, time, time + DEFAULT_NOTE_DURATION, event.value, state.current_velocity,
              is_drum=True)
    elif kind == "velocity":
        bins = vocabularies.num_velocity_bins_from_codec(codec)
        state.current_velocity = vocabularies.bin_to_velocity(event.value, bins)
    elif kind == "program":
        state.current_program = event.value
    elif kind == "tie":
        if not state.is_tie_section:
            raise ValueError("tie section end event when not in tie section")
        for key in list(state.active_pitches):
            if key not in state.tied_pitches:
                onset, vel = state.active_pitches.pop(key)
                _emit(state.note_sequence, onset, state.current_time, key[0], vel, key[1])
        state.is_tie_section = False
    else:
        raise ValueError("unexpected event type: %s" % kind)


def begin_tied_pitches_section(state: NoteDecodingState) -> None:
    state.tied_pitches = set()
    state.is_tie_section = True


def assign_instruments(ns: NoteSequence) -> None:
    """One instrument number per program in order of appearance, skipping 9 (drums) (reference :68-80)."""
    by_program: Dict[int, int] = {}
    for note in ns.notes:
        if note.is_drum:
            note.instrument = 9
        else:
            if note.program not in by_program:
                n = len(by_program)
                by_program[note.program] = n if n < 9 else n + 1
            note.instrument = by_program[note.program]


def flush_note_decoding_state(state: NoteDecodingState) -> NoteSequence:
    """End all still-active notes and return the sequence (reference :381-393)."""
    for onset, _ in state.active_pitches.values():
        state.current_time = max(state.current_time, onset + MIN_NOTE_DURATION)
    for key in list(state.active_pitches):
        onset, vel = state.active_pitches.pop(key)
        _emit(state.note_sequence, onset, state.current_time, key[0], vel, key[1])
    assign_instruments(state.note_sequence)
    return state.note_sequence


@dataclasses.dataclass
class NoteEncodingSpecType:
    init_decoding_state_fn: object
    begin_decoding_segment_fn: object
    decode_event_fn: object
    flush_decoding_state_fn: object


# onsets + offsets with a "tie" section at the start of every segment (the spec inference.py:230 uses)
NoteEncodingWithTiesSpec = NoteEncodingSpecType(NoteDecodingState, begin_tied_pitches_section, decode_note_event,
                                                flush_note_decoding_state)
NoteEncodingSpec = NoteEncodingSpecType(NoteDecodingState, lambda state: None, decode_note_event,
                                        flush_note_decoding_state)


# ---- notes -> timed event data (the tokenisation half; reference :48-66,83-256) ---------------------------
def validate_note_sequence(ns: NoteSequence) -> None:
    """ValueError for an empty-duration or silent note (reference :83-90)."""
    for note in ns.notes:
        if not note.start_time < note.end_time:
            raise ValueError("note has start time >= end time: %f >= %f" % (note.start_time, note.end_time))
        if note.velocity == 0:
            raise ValueError("note has zero velocity")


def trim_overlapping_notes(ns: NoteSequence) -> NoteSequence:
    """Copy of `ns` in which a note that is still sounding when the same (pitch, program, is_drum) starts
    again is cut at that onset; notes left with no duration are dropped (reference :48-66).  The event
    vocabulary cannot express two simultaneous notes of one key."""
    out = NoteSequence([dataclasses.replace(n) for n in ns.notes], ns.total_time, ns.ticks_per_quarter)
    by_key: Dict[Tuple[int, int, bool], List[Note]] = {}
    for note in out.notes:
        by_key.setdefault((note.pitch, note.program, note.is_drum), []).append(note)
    for group in by_key.values():
        group.sort(key=lambda n: n.start_time)              # stable, like sorted()
        for earlier, later in zip(group, group[1:]):
            if earlier.end_time > later.start_time:
                earlier.end_time = later.start_time
    out.notes = [n for n in out.notes if n.start_time < n.end_time]
    return out


def note_arrays_to_note_sequence(onset_times, pitches, offset_times=None, velocities=None, programs=None,
                                 is_drums=None) -> NoteSequence:
    """Parallel arrays -> NoteSequence with the reference's defaults (10 ms, velocity 100, program 0)."""
    ns = NoteSequence()
    for i, (onset, pitch) in enumerate(zip(onset_times, pitches)):
        end = onset + DEFAULT_NOTE_DURATION if offset_times is None else offset_times[i]
        ns.notes.append(Note(onset, end, int(pitch), DEFAULT_VELOCITY if velocities is None else int(velocities[i]),
                             0 if programs is None else int(programs[i]),
                             False if is_drums is None else bool(is_drums[i])))
        ns.total_time = max(ns.total_time, end)
    assign_instruments(ns)
    return ns


@dataclasses.dataclass
class NoteEventData:
    pitch: int
    velocity: int = None          # None: onsets only; 0: note-off
    program: int = None
    is_drum: bool = None
    instrument: int = None


def note_sequence_to_onsets(ns: NoteSequence):
    """(times, values) of note onsets, pitch-sorted so that the later stable time sort breaks ties by pitch."""
    notes = sorted(ns.notes, key=lambda n: n.pitch)
    return [n.start_time for n in notes], [NoteEventData(pitch=n.pitch) for n in notes]


def note_sequence_to_onsets_and_offsets(ns: NoteSequence):
    """All offsets first, then all onsets, each pitch-sorted: equal times put offsets before onsets."""
    notes = sorted(ns.notes, key=lambda n: n.pitch)
    offs = [(n.end_time, NoteEventData(pitch=n.pitch, velocity=0)) for n in notes]
    ons = [(n.start_time, NoteEventData(pitch=n.pitch, velocity=n.velocity)) for n in notes]
    times, values = zip(*(offs + ons)) if notes else ((), ())
    return list(times), list(values)


def note_sequence_to_onsets_and_offsets_and_programs(ns: NoteSequence):
    """As above with programs; drums have no offsets and sort after every pitched program (reference :173-200)."""
    notes = sorted(ns.notes, key=lambda n: (n.is_drum, n.program, n.pitch))
    offs = [(n.end_time, NoteEventData(n.pitch, 0, n.program, False)) for n in notes if not n.is_drum]
    ons = [(n.start_time, NoteEventData(n.pitch, n.velocity, n.program, n.is_drum)) for n in notes]
    both = offs + ons
    return [t for t, _ in both], [v for _, v in both]


@dataclasses.dataclass
class NoteEncodingState:
    """(pitch, program) -> velocity bin of the last event seen for it (0 once released)."""
    active_pitches: Dict[Tuple[int, int], int] = dataclasses.field(default_factory=dict)


def note_event_data_to_events(state, value: NoteEventData, codec: event_codec.Codec):
    """One NoteEventData -> its 1-3 codec events (reference :211-242)."""
    E = event_codec.Event
    if value.velocity is None:
        return [E("pitch", value.pitch)]
    vbin = vocabularies.velocity_to_bin(value.velocity, vocabularies.num_velocity_bins_from_codec(codec))
    if value.program is not None and value.is_drum:
        return [E("velocity", vbin), E("drum", value.pitch)]         # drums: own pitch vocabulary, no program
    program = 0 if value.program is None else value.program
    if state is not None:
        state.active_pitches[(value.pitch, program)] = vbin
    head = [] if value.program is None else [E("program", value.program)]
    return head + [E("velocity", vbin), E("pitch", value.pitch)]


def note_encoding_state_to_events(state: NoteEncodingState):
    """program/pitch pairs of the notes sounding now, ordered by (program, pitch), closed by `tie`."""
    E = event_codec.Event
    out = []
    for pitch, program in sorted(state.active_pitches, key=lambda k: (k[1], k[0])):
        if state.active_pitches[(pitch, program)]:
            out += [E("program", program), E("pitch", pitch)]
    out.append(E("tie", 0))
    return out
