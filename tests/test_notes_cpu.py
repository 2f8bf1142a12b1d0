"""Token post-processing -> notes -> MIDI: the product's restatement against
(a) codec tables recorded from the reference's own contrib/event_codec.py, (b) the oracle's literal
restatement on random token streams, (c) hand-derived known answers for the note state machine."""
import json
import os
import struct

import numpy as np
import pytest

from contrib import event_codec, metrics_utils, midi_io, note_sequences, run_length_encoding, vocabularies
from oracle import notes_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "codec_golden.json")


@pytest.fixture(scope="module")
def codec():
    return vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))


def test_codec_matches_reference_tables(codec):
    g = json.load(open(GOLD))
    assert codec.num_classes == g["num_classes"] == 1388 and codec.max_shift_steps == g["max_shift_steps"]
    assert vocabularies.vocab_size(codec) == 1491            # + 3 specials + 100 extra ids (padded to 1536 in the model)
    for i, (t, v) in g["decode"].items():
        e = codec.decode_event_index(int(i))
        assert (e.type, e.value) == (t, v)
        assert codec.encode_event(e) == int(i)
    for t, r in g["ranges"].items():
        assert list(codec.event_type_range(t)) == r
    for k, idx in g["encode"].items():
        t, v = k.split(":")
        assert codec.encode_event(event_codec.Event(t, int(v))) == idx
    with pytest.raises(ValueError):
        codec.decode_event_index(1388)
    with pytest.raises(ValueError):
        codec.encode_event(event_codec.Event("pitch", 128))
    assert codec.is_shift_event_index(1000) and not codec.is_shift_event_index(1001)


def _tok(codec, *events):
    return np.array([codec.encode_event(event_codec.Event(t, v)) for t, v in events])


def _notes(ns):
    return [(round(n.start_time, 6), round(n.end_time, 6), n.pitch, n.velocity, n.program, n.is_drum, n.instrument)
            for n in ns.notes]


def test_note_state_machine_known_answers(codec):
    spec = note_sequences.NoteEncodingWithTiesSpec
    # segment 0 (start 0.0): empty tie section, program 5 note 60 on at 0.10, drum 36 at 0.30, note off at 0.50,
    # program 40 note 64 on at 0.50 and left sounding
    # (a run of shift tokens encodes the ABSOLUTE time since the segment start: the step counter resets at
    # every non-shift event, run_length_encoding.py:229-236)
    seg0 = _tok(codec, ("tie", 0), ("shift", 10), ("program", 5), ("velocity", 1), ("pitch", 60), ("shift", 30),
                ("drum", 36), ("shift", 25), ("shift", 25), ("velocity", 0), ("pitch", 60), ("program", 40),
                ("velocity", 1), ("pitch", 64))
    # segment 1 (start 2.05 -> 2.05): ties 64/40 over, then ends it at +0.25
    seg1 = _tok(codec, ("program", 40), ("pitch", 64), ("tie", 0), ("shift", 25), ("velocity", 0), ("pitch", 64))
    preds = [{"est_tokens": seg1, "start_time": 2.05, "raw_inputs": []},
             {"est_tokens": seg0, "start_time": 0.0, "raw_inputs": []}]
    res = metrics_utils.event_predictions_to_ns(preds, codec, spec)
    assert res["est_invalid_events"] == 0 and res["est_dropped_events"] == 0 and res["start_times"] == [0.0, 2.05]
    assert _notes(res["est_ns"]) == [
        (0.3, 0.31, 36, 127, 0, True, 9),          # drums: fixed 10 ms, instrument 9
        (0.1, 0.5, 60, 127, 5, False, 0),          # bin_to_velocity(1, 1) = 127
        (0.5, 2.3, 64, 127, 40, False, 1),         # tied across the segment boundary
    ]
    # a note that is NOT re-declared in the next segment's tie section ends at that segment's start
    seg1b = _tok(codec, ("tie", 0), ("shift", 5))
    res = metrics_utils.event_predictions_to_ns([{"est_tokens": seg0, "start_time": 0.0, "raw_inputs": []},
                                                 {"est_tokens": seg1b, "start_time": 2.05, "raw_inputs": []}], codec, spec)
    assert _notes(res["est_ns"])[-1] == (0.5, 2.05, 64, 127, 40, False, 1)
    # invalid events are counted, not fatal: note-off for an inactive pitch, drum at velocity 0, bad index
    bad = np.concatenate([_tok(codec, ("tie", 0), ("velocity", 0), ("pitch", 70), ("drum", 40)), [5000]])
    res = metrics_utils.event_predictions_to_ns([{"est_tokens": bad, "start_time": 0.0, "raw_inputs": []}], codec, spec)
    assert res["est_invalid_events"] == 3 and _notes(res["est_ns"]) == []
    # events beyond the next segment's start are dropped (everything from the offending shift on)
    long0 = _tok(codec, ("tie", 0), ("velocity", 1), ("pitch", 50), ("shift", 300), ("pitch", 51))
    res = metrics_utils.event_predictions_to_ns([{"est_tokens": long0, "start_time": 0.0, "raw_inputs": []},
                                                 {"est_tokens": _tok(codec, ("tie", 0)), "start_time": 2.0, "raw_inputs": []}],
                                                codec, spec)
    assert res["est_dropped_events"] == 2 and _notes(res["est_ns"]) == [(0.0, 2.0, 50, 127, 0, False, 0)]
    # ten programs: instrument numbers skip the drum channel
    many = [("tie", 0), ("velocity", 1)]
    for p in range(11):
        many += [("program", p), ("pitch", 60)]
    res = metrics_utils.event_predictions_to_ns([{"est_tokens": _tok(codec, *many), "start_time": 0.0, "raw_inputs": []}],
                                                codec, spec)
    assert [n.instrument for n in res["est_ns"].notes] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11]


def test_product_equals_oracle_on_random_streams(codec):
    rs = np.random.RandomState(7)
    spec = note_sequences.NoteEncodingWithTiesSpec
    for trial in range(60):
        preds = []
        for s in range(rs.randint(1, 4)):
            n = rs.randint(0, 80)
            kinds = rs.choice(6, size=n, p=[0.3, 0.35, 0.15, 0.03, 0.1, 0.07])
            toks = []
            for k in kinds:
                lo, hi = [(0, 60), (1001, 1128), (1129, 1130), (1131, 1131), (1132, 1140), (1260, 1290)][k]
                toks.append(rs.randint(lo, hi + 1))
            if rs.rand() < 0.2:
                toks.insert(rs.randint(0, len(toks) + 1), 1388 + rs.randint(0, 50))     # invalid id
            preds.append({"est_tokens": np.array(toks, dtype=np.int64), "start_time": 2.05 * s, "raw_inputs": []})
        res = metrics_utils.event_predictions_to_ns(preds, codec, spec)
        ref_notes, inv, drp = notes_ref.predictions_to_notes(preds)
        assert (res["est_invalid_events"], res["est_dropped_events"]) == (inv, drp)
        got = [[n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument] for n in res["est_ns"].notes]
        assert got == ref_notes


def test_to_event_eos_quirk_and_midi_bytes(codec):
    import inference

    class _M:
        class config:
            eos_token_id = 1
        def to(self, *_):
            return self
    h = inference.InferenceHandler.__new__(inference.InferenceHandler)
    h.codec = codec
    seg = np.concatenate([_tok(codec, ("tie", 0), ("velocity", 1), ("program", 3), ("pitch", 72), ("shift", 50),
                               ("velocity", 0), ("pitch", 72)), [-1, -1]])
    no_eos = _tok(codec, ("tie", 0), ("velocity", 1), ("pitch", 40))        # never emitted EOS -> contributes nothing
    ft = [np.array([[0.0, 0.008], [2.048, 2.056]])]
    ns = h._to_event([np.stack([np.pad(seg, (0, 0)), np.pad(no_eos, (0, len(seg) - len(no_eos)), constant_values=5)])], ft)
    ref_notes, _, _ = notes_ref.to_event([np.stack([seg, np.pad(no_eos, (0, len(seg) - len(no_eos)), constant_values=5)])], ft)
    assert _notes(ns) == [(0.0, 0.5, 72, 127, 3, False, 0)]
    assert [[n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument] for n in ns.notes] == ref_notes
    # MIDI: header, tempo track, one instrument track with program change + note on/off at the right ticks
    data = midi_io.note_sequence_to_midi_bytes(ns)
    assert data[:4] == b"MThd" and struct.unpack(">IHHH", data[4:14]) == (6, 1, 2, 220)
    t2 = data.index(b"MTrk", data.index(b"MTrk") + 4)
    body = data[t2 + 8:]
    assert body[:3] == bytes([0x00, 0xC0, 3])                       # delta 0, program change 3 on channel 0
    assert body[3:7] == bytes([0x00, 0x90, 72, 127])                # note on at tick 0
    # 0.5 s at 120 bpm, 220 tpq = 220 ticks -> VLQ 0x81 0x5C
    assert body[7:12] == bytes([0x81, 0x5C, 0x80, 72, 0])
    assert data.endswith(b"\x00\xff\x2f\x00")
