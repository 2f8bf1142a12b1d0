"""Target tokenisation (mrmt3.tokenizer + the encode halves of contrib.note_sequences /
contrib.run_length_encoding) against the literal restatement in oracle/notes_ref.py, known answers, and the
round trip through the (separately checked) decoder."""
import numpy as np
import pytest

from contrib import metrics_utils, note_sequences as nsq, vocabularies
from contrib.note_sequences import Note, NoteSequence
from mrmt3.tokenizer import Tokenizer
from oracle import notes_ref as ref


def _random_notes(seed, n=60, dur=12.0, drums=True):
    rs = np.random.RandomState(seed)
    notes = []
    for _ in range(n):
        s = float(rs.uniform(0, dur - 0.5))
        e = s + float(rs.uniform(0.03, 1.5))
        is_drum = drums and rs.rand() < 0.2
        notes.append(Note(round(s, 3), round(e, 3), int(rs.randint(30, 90)), int(rs.randint(1, 128)),
                          0 if is_drum else int(rs.choice([0, 25, 33, 48])), bool(is_drum)))
    return NoteSequence(notes, max(n.end_time for n in notes))


def _as_lists(ns):
    return [[n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum] for n in ns.notes]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tokenize_matches_oracle(seed):
    tk = Tokenizer()
    ns = _random_notes(seed)
    n_samples = int(12.0 * 16000) + 77
    feats = tk.tokenize(ns, n_samples)
    notes = ref.trim_overlapping(_as_lists(ns))
    times, values = ref.onsets_offsets_programs(notes)
    ft = np.arange((n_samples + 128 - n_samples % 128) // 128) / 125.0
    want = ref.encode_and_index(times, values, ft)
    got = (feats["targets"], feats["input_event_start_indices"], feats["input_event_end_indices"],
           feats["state_events"], feats["input_state_event_indices"])
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    np.testing.assert_array_equal(feats["input_times"], ft)
    # every frame's slice ends where the next begins; rows of 256 frames
    assert (feats["input_event_end_indices"][:-1] == feats["input_event_start_indices"][1:]).all()
    for start in (0, 256, 700, len(ft) - 256):
        row = tk.extract_target_sequence(feats, start, 256)
        np.testing.assert_array_equal(row, ref.extract_targets(want, start, 256))
        np.testing.assert_array_equal(tk.run_length_encode_shifts(row), ref.rle_shifts(row))


def test_known_answers():
    tk = Tokenizer()
    c = tk.codec
    P, V, T = c.event_type_range("program")[0], c.event_type_range("velocity")[0], tk.tie_token
    pitch0 = c.event_type_range("pitch")[0]
    # one piano note 0.50 s .. 1.00 s, pitch 60, in a 2.048 s recording (256 frames + the pad frame)
    ns = NoteSequence([Note(0.5, 1.0, 60, 100, 0, False)], 1.0)
    feats = tk.tokenize(ns, 32768)
    ev = feats["targets"]
    assert len(feats["input_times"]) == 257
    # 50 single shifts, onset (program 0, velocity 1, pitch 60), 50 shifts, offset (program 0, velocity 0, pitch 60),
    # then shifts up to and including the step on the last frame time (256/125 = 2.048 s -> step 205)
    assert (ev[:50] == 1).all() and ev[50:53].tolist() == [P, V + 1, pitch0 + 60]
    assert (ev[53:103] == 1).all() and ev[103:106].tolist() == [P, V, pitch0 + 60]
    assert (ev[106:] == 1).all() and len(ev) == 6 + 205
    # state dumps: before the onset nothing sounds -> [tie]; before the offset the note sounds -> [program, pitch, tie]
    assert feats["state_events"].tolist() == [T, P, pitch0 + 60, T]
    # whole-recording row: the tie section is empty, shifts are absolute and the trailing ones are dropped
    row = tk.run_length_encode_shifts(tk.extract_target_sequence(feats, 0, 256))
    assert row.tolist() == [T, 50, P, V + 1, pitch0 + 60, 100, V, pitch0 + 60]      # 2nd `program 0` is a repeat
    # a row starting at frame 100 (0.8 s) finds the note sounding: it is declared before the tie token
    row = tk.run_length_encode_shifts(tk.extract_target_sequence(feats, 100, 100))
    assert row.tolist() == [P, pitch0 + 60, T, 20, V, pitch0 + 60]                  # offset 0.2 s into the row
    # shifts longer than max_shift_steps are split, and the value is the TOTAL since the row start
    many = np.concatenate([np.ones(1500, np.int64), [pitch0 + 1], np.ones(10, np.int64), [pitch0 + 2]])
    assert tk.run_length_encode_shifts(many).tolist() == [1000, 500, pitch0 + 1, 1000, 510, pitch0 + 2]
    # repeat filter alone
    assert tk.remove_redundant_tokens([P, V + 1, 5, P, V + 1, 6, P + 1, 7]).tolist() == [P, V + 1, 5, 6, P + 1, 7]


def test_trim_overlapping_and_validate():
    ns = NoteSequence([Note(0.0, 1.0, 60, 90), Note(0.5, 0.8, 60, 90), Note(0.5, 0.5 + 1e-9, 61, 90),
                       Note(0.2, 0.4, 60, 90, 0, True)], 1.0)
    out = nsq.trim_overlapping_notes(ns)
    assert [(n.start_time, n.end_time, n.pitch, n.is_drum) for n in out.notes][:2] == [(0.0, 0.5, 60, False), (0.5, 0.8, 60, False)]
    assert ns.notes[0].end_time == 1.0                       # the input is not modified
    with pytest.raises(ValueError):
        nsq.validate_note_sequence(NoteSequence([Note(1.0, 1.0, 60, 90)]))
    with pytest.raises(ValueError):
        nsq.validate_note_sequence(NoteSequence([Note(0.0, 1.0, 60, 0)]))


@pytest.mark.parametrize("seed", [3, 4])
def test_round_trip_through_the_decoder(seed):
    """tokenise -> rows of 256 frames -> decode with ties == the (trimmed, 10 ms-quantised) notes."""
    tk = Tokenizer()
    ns = nsq.trim_overlapping_notes(_random_notes(seed, n=40, dur=10.0, drums=False))
    n_samples = int(11.0 * 16000)
    feats = tk.tokenize(ns, n_samples)
    n_frames = len(feats["input_times"])
    preds = []
    for start in range(0, n_frames, 256):
        n = min(256, n_frames - start)
        row = tk.run_length_encode_shifts(tk.extract_target_sequence(feats, start, n))
        t0 = feats["input_times"][start]
        preds.append({"est_tokens": row, "start_time": t0 - t0 % 0.01, "raw_inputs": []})
    got = metrics_utils.event_predictions_to_ns(preds, codec=tk.codec, encoding_spec=nsq.NoteEncodingWithTiesSpec)["est_ns"]
    q = lambda t: round(t * 100)
    want = sorted((q(n.start_time), max(q(n.end_time), q(n.start_time) + 1), n.pitch, n.program) for n in ns.notes
                  if q(n.start_time) != q(n.end_time) or True)
    have = sorted((q(n.start_time), q(n.end_time), n.pitch, n.program) for n in got.notes)
    # onsets, pitches and programs survive exactly; offsets to the 10 ms grid
    assert [w[0::2] for w in want] == [h[0::2] for h in have]
    assert sum(abs(w[1] - h[1]) for w, h in zip(want, have)) == 0


def test_token_order_augmentation_matches_reference_procedure():
    """is_randomize_tokens: rows keep every program / velocity token, note groups between shifts are shuffled with the
    same np.random draws as the reference's name-based procedure, repeats are dropped afterwards."""
    tk = Tokenizer(is_randomize_tokens=True)
    ns = _random_notes(7, n=80, dur=6.0)
    feats = tk.tokenize(ns, int(6.5 * 16000))
    plain = tk.run_length_encode_shifts(tk.extract_target_sequence(feats, 0, 256))
    names = [ref.token_name(t) for t in plain]
    want = ref.remove_redundant(np.array([ref.token_index(n) for n in ref.randomize_tokens(list(names), np.random.RandomState(11))]))
    got = tk.row_targets(feats, 0, 256, rng=np.random.RandomState(11))
    np.testing.assert_array_equal(got, want)
    # the shuffle permutes groups inside a span: the multiset of (velocity?, pitch/drum) tokens per span is unchanged
    shuffled = tk.randomize_tokens(plain, np.random.RandomState(3))
    assert sorted(shuffled.tolist()) == sorted(plain.tolist()) and not np.array_equal(shuffled, plain)
    # everything up to the first shift (the tie section) and from the last shift on is untouched
    first = next(i for i, t in enumerate(plain) if t < 1000)
    last = max(i for i, t in enumerate(plain) if t < 1000)
    np.testing.assert_array_equal(shuffled[:first + 1], plain[:first + 1])
    np.testing.assert_array_equal(shuffled[last:], plain[last:])
    # names <-> ids
    assert [ref.token_index(ref.token_name(i)) for i in (0, 5, 999, 1001, 1128, 1129, 1130, 1131, 1132, 1259, 1260, 1387)] == \
        [0, 5, 999, 1001, 1128, 1129, 1130, 1131, 1132, 1259, 1260, 1387]
