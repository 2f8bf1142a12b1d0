"""Minimal Standard-MIDI-File writer for the decoded notes — the build's replacement for
`note_seq.sequence_proto_to_midi_file` (called at inference.py:201).  Format 1, 220 ticks per quarter
at 120 bpm (note_seq's defaults), one track per instrument, drums on channel 10.
"""
from __future__ import annotations

import struct
from typing import List

from contrib.note_sequences import NoteSequence

QPM = 120.0


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _track(events) -> bytes:
    body, last = bytearray(), 0
    for tick, _, data in sorted(events, key=lambda e: (e[0], e[1])):
        body += _vlq(tick - last) + data
        last = tick
    body += b"\x00\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(body)) + bytes(body)


def note_sequence_to_midi_bytes(ns: NoteSequence) -> bytes:
    tpq = ns.ticks_per_quarter
    to_tick = lambda t: int(round(t * QPM / 60.0 * tpq))
    tracks: List[bytes] = [_track([(0, 0, b"\xff\x51\x03" + struct.pack(">I", int(60e6 / QPM))[1:])])]
    by_inst = {}
    for n in ns.notes:
        by_inst.setdefault((n.instrument, n.program, n.is_drum), []).append(n)
    next_ch = 0
    for (inst, program, is_drum), notes in sorted(by_inst.items()):
        if is_drum:
            ch = 9
        else:
            ch = next_ch % 16
            if ch == 9:
                next_ch += 1
                ch = next_ch % 16
            next_ch += 1
        ev = [(0, 0, bytes([0xC0 | ch, program & 0x7F]))]
        for n in notes:
            ev.append((to_tick(n.start_time), 2, bytes([0x90 | ch, n.pitch & 0x7F, max(1, n.velocity & 0x7F)])))
            ev.append((max(to_tick(n.end_time), to_tick(n.start_time) + 1), 1, bytes([0x80 | ch, n.pitch & 0x7F, 0])))
        tracks.append(_track(ev))
    return b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), tpq) + b"".join(tracks)


def note_sequence_to_midi_file(ns: NoteSequence, path: str) -> None:
    with open(path, "wb") as f:
        f.write(note_sequence_to_midi_bytes(ns))


# ---- reader (replaces pretty_midi.PrettyMIDI / note_seq.midi_file_to_note_sequence for evaluate.py) ----------
import dataclasses


@dataclasses.dataclass
class MidiNote:
    start: float
    end: float
    pitch: int
    velocity: int


@dataclasses.dataclass
class MidiInstrument:
    program: int
    is_drum: bool
    notes: list
    track: int = 0
    channel: int = 0


@dataclasses.dataclass
class MidiData:
    instruments: list
    resolution: int
    end_time: float


def _read_vlq(buf, pos):
    value = 0
    while True:
        byte = buf[pos]
        pos += 1
        value = (value << 7) | (byte & 0x7F)
        if not byte & 0x80:
            return value, pos


def _parse_track(buf):
    """-> list of (abs_tick, kind, channel, a, b); kinds: 'on', 'off', 'program', 'tempo'."""
    events, pos, tick, status = [], 0, 0, None
    n = len(buf)
    while pos < n:
        delta, pos = _read_vlq(buf, pos)
        tick += delta
        byte = buf[pos]
        if byte == 0xFF:                                   # meta
            kind = buf[pos + 1]
            length, p2 = _read_vlq(buf, pos + 2)
            data = buf[p2:p2 + length]
            pos = p2 + length
            if kind == 0x51 and length == 3:
                events.append((tick, "tempo", 0, int.from_bytes(data, "big"), 0))
            elif kind == 0x2F:
                break
            continue
        if byte in (0xF0, 0xF7):                           # sysex
            length, p2 = _read_vlq(buf, pos + 1)
            pos = p2 + length
            continue
        if byte & 0x80:
            status = byte
            pos += 1
        elif status is None:
            raise ValueError("MIDI data byte without a running status")
        hi, ch = status & 0xF0, status & 0x0F
        if hi in (0xC0, 0xD0):                             # one data byte
            a = buf[pos]
            pos += 1
            if hi == 0xC0:
                events.append((tick, "program", ch, a, 0))
        else:                                              # two data bytes
            a, b = buf[pos], buf[pos + 1]
            pos += 2
            if hi == 0x90 and b > 0:
                events.append((tick, "on", ch, a, b))
            elif hi == 0x80 or hi == 0x90:
                events.append((tick, "off", ch, a, b))
    return events


def read_midi(source) -> MidiData:
    """Standard MIDI File (format 0/1, metrical time) -> instruments with notes in seconds, following
    pretty_midi's conventions: tempo changes are taken from track 0 only; one instrument per
    (program, channel, track); channel 10 is drums; a note-off closes every open note of its key that
    started on an earlier tick (a note-on and note-off on the same tick produce nothing); the program of a
    note is the one in force on its channel when it ENDS."""
    buf = source if isinstance(source, (bytes, bytearray)) else open(source, "rb").read()
    if buf[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File")
    hlen, fmt, ntrks, division = struct.unpack(">IHHH", buf[4:14])
    if division & 0x8000:
        raise ValueError("SMPTE time division is not supported")
    pos = 8 + hlen
    tracks = []
    for _ in range(ntrks):
        if buf[pos:pos + 4] != b"MTrk":
            raise ValueError("bad track chunk")
        length = struct.unpack(">I", buf[pos + 4:pos + 8])[0]
        tracks.append(_parse_track(buf[pos + 8:pos + 8 + length]))
        pos += 8 + length
    # tempo map from track 0: piecewise-linear tick -> seconds
    tempo_pts = [(0, 0.0, 500000)]
    for tick, kind, _, us, _ in (tracks[0] if tracks else []):
        if kind != "tempo":
            continue
        t0, s0, cur = tempo_pts[-1]
        if tick == t0:
            tempo_pts[-1] = (t0, s0, us)
        else:
            tempo_pts.append((tick, s0 + (tick - t0) * cur / 1e6 / division, us))

    def to_time(tick):
        lo, hi = 0, len(tempo_pts) - 1
        while lo < hi:                                     # last point with point.tick <= tick
            mid = (lo + hi + 1) // 2
            if tempo_pts[mid][0] <= tick:
                lo = mid
            else:
                hi = mid - 1
        t0, s0, us = tempo_pts[lo]
        return s0 + (tick - t0) * us / 1e6 / division

    instruments, index, end_time = [], {}, 0.0
    for ti, events in enumerate(tracks):
        program = [0] * 16
        open_notes = {}
        for tick, kind, ch, a, b in events:
            if kind == "program":
                program[ch] = a
            elif kind == "on":
                open_notes.setdefault((ch, a), []).append((tick, b))
            elif kind == "off" and (ch, a) in open_notes:
                pending = open_notes[(ch, a)]
                closing = [(s, v) for s, v in pending if s != tick]
                keeping = [(s, v) for s, v in pending if s == tick]
                for s, v in closing:
                    key = (program[ch], ch, ti)
                    if key not in index:
                        index[key] = MidiInstrument(program[ch], ch == 9, [], ti, ch)
                        instruments.append(index[key])
                    index[key].notes.append(MidiNote(to_time(s), to_time(tick), a, v))
                    end_time = max(end_time, to_time(tick))
                if closing and keeping:
                    open_notes[(ch, a)] = keeping
                else:
                    del open_notes[(ch, a)]
    return MidiData(instruments, division, end_time)


def midi_to_note_sequence(midi: MidiData) -> NoteSequence:
    from contrib.note_sequences import Note
    ns = NoteSequence(ticks_per_quarter=midi.resolution)
    for k, inst in enumerate(midi.instruments):
        for n in inst.notes:
            ns.notes.append(Note(n.start, n.end, n.pitch, n.velocity, inst.program, inst.is_drum, k))
            ns.total_time = max(ns.total_time, n.end)
    return ns


def midi_file_to_note_sequence(source) -> NoteSequence:
    return midi_to_note_sequence(read_midi(source))
