"""evaluate.py: the MIDI reader (pretty_midi's conventions), the note matcher (mir_eval's
algorithm; mir_eval itself is absent, so: maximum matching checked against scipy's, scores against
hand-derived cases) and the program-aware scores."""
import struct

import numpy as np
import pytest

import evaluate
from contrib import midi_io
from contrib import transcription_metrics as tm
from contrib.note_sequences import Note, NoteSequence


def test_maximum_matching_agrees_with_scipy():
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    rs = np.random.RandomState(0)
    for trial in range(60):
        nl, nr = rs.randint(1, 25), rs.randint(1, 25)
        dense = rs.rand(nl, nr) < rs.choice([0.05, 0.15, 0.4])
        adj = {u: list(np.nonzero(dense[u])[0]) for u in range(nl) if dense[u].any()}
        m = tm._max_bipartite_matching(adj, nl)
        assert len(set(m.values())) == len(m) and all(dense[u, v] for u, v in m.items())
        want = (maximum_bipartite_matching(csr_matrix(dense), perm_type="column") >= 0).sum()
        assert len(m) == want


def test_onset_scores_known_answers():
    hz = tm.midi_to_hz
    assert abs(hz(69) - 440.0) < 1e-9 and abs(hz(81) - 880.0) < 1e-9
    ref_iv = np.array([[0.0, 1.0], [1.0, 2.0], [2.0, 3.0], [3.0, 4.0]])
    ref_p = hz([60, 62, 64, 65])
    # hit (30 ms late), hit exactly on the 50 ms boundary (<=), miss by pitch (a semitone), miss by onset, extra note
    est_iv = np.array([[0.03, 0.5], [1.05, 1.2], [2.0, 3.0], [3.06, 4.0], [5.0, 5.5]])
    est_p = hz([60, 62, 65, 65, 70])
    p, r, f, ov = tm.precision_recall_f1_overlap(ref_iv, ref_p, est_iv, est_p, offset_ratio=None)
    assert (p, r) == (2 / 5, 2 / 4) and abs(f - 2 * p * r / (p + r)) < 1e-12
    # with offsets: note 0's offset is 0.5 s early (tolerance max(0.2*1.0, 0.05) = 0.2) -> only... note 1 is 0.8 early -> none
    p2, r2, f2, _ = tm.precision_recall_f1_overlap(ref_iv, ref_p, est_iv, est_p)
    assert (p2, r2, f2) == (0.0, 0.0, 0.0)
    # overlap ratio of the matched pairs: (0.5-0.03)/(1.0-0.0) and (1.2-1.05)/(2.0-1.0)
    assert abs(ov - np.mean([0.47 / 1.0, 0.15 / 1.0])) < 1e-12
    # one estimated note cannot serve two reference notes
    p, r, f, _ = tm.precision_recall_f1_overlap(np.array([[0.0, 1.0], [0.02, 1.0]]), hz([60, 60]),
                                                np.array([[0.01, 1.0]]), hz([60]), offset_ratio=None)
    assert (p, r) == (1.0, 0.5)
    # empty sides and invalid input
    assert tm.precision_recall_f1_overlap(np.zeros((0, 2)), np.zeros(0), est_iv, est_p) == (0.0, 0.0, 0.0, 0.0)
    with pytest.raises(ValueError):
        tm.precision_recall_f1_overlap(np.array([[1.0, 1.0]]), hz([60]), est_iv, est_p)
    assert tm.f_measure(0, 0) == 0.0 and tm.f_measure(1.0, 0.5) == pytest.approx(2 / 3)


def _smf(tracks, division=480, fmt=1):
    out = b"MThd" + struct.pack(">IHHH", 6, fmt, len(tracks), division)
    for body in tracks:
        out += b"MTrk" + struct.pack(">I", len(body)) + body
    return out


def test_midi_reader_conventions():
    tempo = b"\x00\xff\x51\x03\x07\xa1\x20" + b"\x83\x60\xff\x51\x03\x03\xd0\x90" + b"\x00\xff\x2f\x00"   # 120 bpm, then 240 bpm at tick 480
    trk = (b"\x00\xc0\x19"            # program 25 on channel 0
           b"\x00\x90\x3c\x64"        # t=0    note on 60
           b"\x83\x60\x3e\x50"        # t=480  note on 62 (running status)
           b"\x83\x60\x80\x3c\x00"    # t=960  note off 60
           b"\x00\x90\x3e\x00"        # t=960  note on 62 velocity 0 = note off
           b"\x00\x99\x24\x7f"        # t=960  drum (channel 10) note on 36
           b"\x00\x89\x24\x00"        # t=960  ... and off on the same tick: produces no note
           b"\x00\xc0\x30"            # program change to 48 while nothing sounds
           b"\x00\x90\x40\x40\x81\x70\x80\x40\x00"   # t=960..1200 note 64 under program 48
           b"\x00\xff\x2f\x00")
    midi = midi_io.read_midi(_smf([tempo, trk]))
    by_prog = {(i.program, i.is_drum): i for i in midi.instruments}
    assert set(by_prog) == {(25, False), (48, False)}
    n60, n62 = by_prog[(25, False)].notes
    assert (n60.pitch, n60.start, n60.end, n60.velocity) == (60, 0.0, 0.75, 100)      # 480 ticks at 120 bpm + 480 at 240 bpm
    assert (n62.pitch, n62.start, n62.end) == (62, 0.5, 0.75)
    n64 = by_prog[(48, False)].notes[0]
    assert (n64.start, n64.end) == (0.75, 0.75 + 240 / 480 * 0.25)
    ns = midi_io.midi_to_note_sequence(midi)
    assert len(ns.notes) == 3 and ns.total_time == n64.end
    with pytest.raises(ValueError):
        midi_io.read_midi(b"RIFF....")


def test_writer_reader_round_trip(tmp_path):
    ns = NoteSequence([Note(0.10, 0.50, 60, 90, 0, False, 0), Note(0.25, 1.00, 64, 70, 33, False, 1),
                       Note(0.50, 0.60, 38, 127, 0, True, 9)], 1.0)
    path = tmp_path / "a.mid"
    midi_io.note_sequence_to_midi_file(ns, str(path))
    back = midi_io.midi_file_to_note_sequence(str(path))
    got = sorted((round(n.start_time, 3), round(n.end_time, 3), n.pitch, n.velocity, n.program, n.is_drum) for n in back.notes)
    want = sorted((n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in ns.notes)
    assert [g[2:] for g in got] == [w[2:] for w in want]
    assert np.allclose([g[:2] for g in got], [w[:2] for w in want], atol=60 / 120 / 220)      # one tick


def test_program_aware_scores(tmp_path):
    ref = NoteSequence([Note(0.0, 0.5, 60, 90, 0, False, 0), Note(1.0, 1.5, 62, 90, 0, False, 0),
                        Note(0.0, 0.5, 40, 90, 33, False, 1), Note(1.0, 1.1, 36, 90, 0, True, 9)], 1.5)
    # the bass note is transcribed with the wrong program (34: same family as 33), one piano note is missing,
    # the drum hit is right, one spurious piano note
    est = NoteSequence([Note(0.01, 0.4, 60, 80, 0, False, 0), Note(0.0, 0.5, 40, 80, 34, False, 1),
                        Note(1.0, 1.1, 36, 80, 0, True, 9), Note(2.0, 2.2, 72, 80, 0, False, 0)], 2.2)
    rp, ep = str(tmp_path / "ref.mid"), str(tmp_path / "est.mid")
    midi_io.note_sequence_to_midi_file(ref, rp)
    midi_io.note_sequence_to_midi_file(est, ep)
    flat = evaluate.mt3_program_aware_note_scores(rp, ep, "flat")
    full = evaluate.mt3_program_aware_note_scores(rp, ep, "full")
    cls = evaluate.mt3_program_aware_note_scores(rp, ep, "midi_class")
    # instrument-agnostic onsets: 3 of 4 estimated and 3 of 4 reference notes match
    assert flat["Onset precision"] == 0.75 and flat["Onset recall"] == 0.75 and flat["Onset F1"] == 0.75
    # flat = pitched vs drums: same 3 matches
    assert flat["Onset + program F1 (flat)"] == pytest.approx(0.75)
    # full program numbers: the bass note no longer matches (33 vs 34) -> 2 of 4 each way
    assert full["Onset + program precision (full)"] == pytest.approx(0.5)
    assert full["Onset + program recall (full)"] == pytest.approx(0.5)
    # MIDI class: 33 and 34 are both "bass" -> back to 3 matches; per-family F1 reported
    assert cls["Onset + program F1 (midi_class)"] == pytest.approx(0.75)
    assert cls["F1 by program"][32] == 1.0 and cls["F1 by program"][-1] == 1.0 and cls["F1 by program"][0] == pytest.approx(0.5)
    m = evaluate.compute_transcription_metrics(rp, ep)
    assert m["len_ref_intervals"] == 4 and m["on_f1"] == 0.75
    score, n_ref, n_est = evaluate.loop_transcription_eval(rp, ep)
    assert (n_ref, n_est) == (3, 3) and 0.0 < score <= 1.0
    assert evaluate.get_granular_program(57, False, "midi_class") == 56 and evaluate.get_granular_program(5, True, "flat") == 1


def test_evaluate_main_directory_layout(tmp_path, capsys):
    ns = NoteSequence([Note(0.0, 0.5, 60, 90, 0, False, 0)], 0.5)
    for t in ("Track01", "Track02"):
        (tmp_path / "est" / t).mkdir(parents=True)
        (tmp_path / "gt" / t).mkdir(parents=True)
        midi_io.note_sequence_to_midi_file(ns, str(tmp_path / "est" / t / "mix.mid"))
        midi_io.note_sequence_to_midi_file(ns, str(tmp_path / "gt" / t / "all_src_v2.mid"))
    scores = evaluate.evaluate_main("Slakh", str(tmp_path / "est"), str(tmp_path / "gt"), enable_instrument_eval=True)
    assert scores["Onset F1"] == 1.0 and scores["Onset + program F1 (full)"] == 1.0
    assert scores["F1 by program"] == {0: 1.0}
    with pytest.raises(ValueError):
        evaluate.evaluate_main("Other", "a", "b")


def test_scores_equal_brute_force_optimal_matching_on_random_small_cases():
    """contrib/transcription_metrics.py restates mir_eval's note matching (absent here: unpinned).  Property check
    against an exhaustive search: for random small note sets the number of matched pairs equals the size of the LARGEST
    one-to-one assignment among the pairs that satisfy the onset / pitch / offset rules, so precision / recall / F1 are
    the optimum mir_eval defines (its matching is a maximum bipartite matching)."""
    import itertools
    from contrib import transcription_metrics as tm
    rs = np.random.RandomState(5)
    for case in range(60):
        nr, ne = rs.randint(1, 7), rs.randint(1, 7)
        ref_on = np.sort(rs.rand(nr) * 2.0)
        ref = np.stack([ref_on, ref_on + 0.1 + rs.rand(nr) * 0.5], 1)
        est_on = np.clip(ref_on[rs.randint(0, nr, ne)] + rs.randn(ne) * 0.04, 0, None)
        est = np.stack([est_on, est_on + 0.1 + rs.rand(ne) * 0.5], 1)
        rp = tm.midi_to_hz(rs.randint(60, 64, nr).astype(float))
        ep = tm.midi_to_hz(rs.randint(60, 64, ne).astype(float))
        for offset_ratio in (None, 0.2):
            p, r, f, _ = tm.precision_recall_f1_overlap(ref, rp, est, ep, offset_ratio=offset_ratio)
            ok = np.abs(np.subtract.outer(ref[:, 0], est[:, 0])).round(4) <= 0.05
            ok &= np.abs(1200 * np.subtract.outer(np.log2(rp), np.log2(ep))) <= 50.0
            if offset_ratio is not None:
                tol = np.maximum(offset_ratio * (ref[:, 1] - ref[:, 0]), 0.05)
                ok &= np.abs(np.subtract.outer(ref[:, 1], est[:, 1])).round(4) <= tol[:, None]
            best = 0
            small, big, mat = (nr, ne, ok) if nr <= ne else (ne, nr, ok.T)
            for perm in itertools.permutations(range(big), small):
                best = max(best, sum(mat[i, j] for i, j in enumerate(perm)))
                if best == small:
                    break
            assert abs(p - best / ne) < 1e-12 and abs(r - best / nr) < 1e-12, (case, offset_ratio, p, r, best)
