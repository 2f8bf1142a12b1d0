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
