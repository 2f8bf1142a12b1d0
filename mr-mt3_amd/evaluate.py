"""`evaluate` — multi-track transcription scores for the MIDI files `test.py` writes, the build's
counterpart of the reference's evaluate.py:16-330 without pretty_midi / note_seq / mir_eval / librosa:
MIDI files are read by `contrib.midi_io.read_midi` (pretty_midi's conventions), note matching is
`contrib.transcription_metrics` (mir_eval's algorithm).  Function names, arguments and result keys are
the reference's.  CPU, host-side: nothing here touches the GPU path.
"""
from __future__ import annotations

import collections
import concurrent.futures
import glob

import numpy as np

from contrib import midi_io
from contrib import transcription_metrics as tm


def get_granular_program(program_number, is_drum, granularity_type):
    """evaluate.py:16-22 ('flat': pitched vs drums; 'midi_class': the General-MIDI family of 8)."""
    if granularity_type == "full":
        return program_number
    if granularity_type == "midi_class":
        return (program_number // 8) * 8
    if granularity_type == "flat":
        return 0 if not is_drum else 1
    return None


def _midi(x):
    return x if isinstance(x, midi_io.MidiData) else midi_io.read_midi(x)


def compute_transcription_metrics(ref_mid, est_mid):
    """evaluate.py:25-53: onset+offset and onset-only scores over all notes (pitches are passed to the matcher
    as MIDI numbers, exactly as the reference does)."""
    ns_ref = midi_io.midi_to_note_sequence(_midi(ref_mid))
    ns_est = midi_io.midi_to_note_sequence(_midi(est_mid))
    iv_r, p_r, _ = tm.sequence_to_valued_intervals(ns_ref)
    iv_e, p_e, _ = tm.sequence_to_valued_intervals(ns_est)
    onoff = tm.precision_recall_f1_overlap(iv_r, p_r, iv_e, p_e)
    on = tm.precision_recall_f1_overlap(iv_r, p_r, iv_e, p_e, offset_ratio=None)
    keys = ("precision", "recall", "f1", "overlap")
    out = {"len_ref_intervals": len(iv_r), "len_est_intervals": len(iv_e)}
    out.update({f"onoff_{k}": v for k, v in zip(keys, onoff)})
    out.update({f"on_{k}": v for k, v in zip(keys, on)})
    return out


def mt3_program_aware_note_scores(fname1, fname2, granularity_type):
    """evaluate.py:56-237: instrument-agnostic onset F1 plus the program-aware onset F1 of MT3 at the given
    program granularity (per-group scores weighted by note counts)."""
    ref_mid, est_mid = _midi(fname1), _midi(fname2)
    res = {}
    iv_r, p_r, _ = tm.sequence_to_valued_intervals(midi_io.midi_to_note_sequence(ref_mid))
    iv_e, p_e, _ = tm.sequence_to_valued_intervals(midi_io.midi_to_note_sequence(est_mid))
    precision, recall, f1, _ = tm.precision_recall_f1_overlap(iv_r, p_r, iv_e, p_e, offset_ratio=None)
    res["Onset precision"], res["Onset recall"], res["Onset F1"] = precision, recall, f1

    def group(mid):
        g = collections.OrderedDict()
        for inst in mid.instruments:
            key = (get_granular_program(inst.program, inst.is_drum, granularity_type), inst.is_drum)
            g.setdefault(key, []).extend(inst.notes)
        return g

    ref_g, est_g = group(ref_mid), group(est_mid)
    sums = {True: [0.0, 0, 0.0, 0], False: [0.0, 0, 0.0, 0]}       # is_drum -> [P*n_est, n_est, R*n_ref, n_ref]
    program_f1 = {}
    for key in set(ref_g) | set(est_g):
        program, is_drum = key

        def arrays(notes):
            iv = np.array([[n.start, n.end] for n in notes], dtype=np.float64).reshape(-1, 2)
            return iv, tm.midi_to_hz([n.pitch for n in notes])

        r_iv, r_hz = arrays(ref_g.get(key, []))
        e_iv, e_hz = arrays(est_g.get(key, []))
        p, r, f, _ = tm.precision_recall_f1_overlap(r_iv, r_hz, e_iv, e_hz, offset_ratio=None)
        if granularity_type == "midi_class":
            program_f1[-1 if is_drum else program] = f
        acc = sums[bool(is_drum)]
        acc[0] += p * len(e_iv)
        acc[1] += len(e_iv)
        acc[2] += r * len(r_iv)
        acc[3] += len(r_iv)
    p_sum, p_cnt = sums[True][0] + sums[False][0], sums[True][1] + sums[False][1]
    r_sum, r_cnt = sums[True][2] + sums[False][2], sums[True][3] + sums[False][3]
    precision = p_sum / p_cnt if p_cnt else 0
    recall = r_sum / r_cnt if r_cnt else 0
    res.update({f"Onset + program precision ({granularity_type})": precision,
                f"Onset + program recall ({granularity_type})": recall,
                f"Onset + program F1 ({granularity_type})": tm.f_measure(precision, recall),
                "F1 by program": program_f1})
    return res


def loop_transcription_eval(ref_mid, est_mid):
    """evaluate.py:240-272: best onset+offset F1 over estimated tracks for every reference track, averaged."""
    ref_mid, est_mid = _midi(ref_mid), _midi(est_mid)
    score = np.zeros((len(ref_mid.instruments), len(est_mid.instruments)))
    for i, r in enumerate(ref_mid.instruments):
        for j, e in enumerate(est_mid.instruments):
            if r.is_drum != e.is_drum:
                continue
            r_iv = np.array([[n.start, n.end] for n in r.notes]).reshape(-1, 2)
            e_iv = np.array([[n.start, n.end] for n in e.notes]).reshape(-1, 2)
            score[i, j] = tm.precision_recall_f1_overlap(r_iv, tm.midi_to_hz([n.pitch for n in r.notes]), e_iv,
                                                         tm.midi_to_hz([n.pitch for n in e.notes]))[2]
    return float(np.mean(np.max(score, axis=-1))), len(ref_mid.instruments), len(est_mid.instruments)


def evaluate_main(dataset_name, test_midi_dir, ground_truth_midi_dir, enable_instrument_eval=False, first_n=None):
    """evaluate.py:275-330: pair every transcribed file with its ground truth, score at the three program
    granularities, print and return the means."""
    if dataset_name == "Slakh":
        est = sorted(glob.glob(f"{test_midi_dir}/*/mix.mid"))
        ref = [k.replace(test_midi_dir, ground_truth_midi_dir).replace("/mix.mid", "/all_src_v2.mid") for k in est]
        if first_n:
            est, ref = est[:first_n], ref[:first_n]
    elif dataset_name in ("ComMU", "NSynth"):
        est = sorted(glob.glob(f"{test_midi_dir}/*.mid"))
        ref = [k.replace(test_midi_dir, ground_truth_midi_dir).replace("_16k.mid", ".mid") for k in est]
    else:
        raise ValueError("dataset_name must be either Slakh or ComMU")

    def score(pair):
        out = {}
        for granularity in ("flat", "full", "midi_class"):
            out.update(mt3_program_aware_note_scores(pair[0], pair[1], granularity))
        return out

    scores = collections.defaultdict(list)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        for fut in concurrent.futures.as_completed([pool.submit(score, p) for p in zip(ref, est)]):
            try:
                for k, v in fut.result().items():
                    scores[k].append(v)
            except Exception as exc:                      # one unreadable file must not end the run
                print(str(exc))
    mean_scores = {k: float(np.mean(v)) for k, v in scores.items() if k != "F1 by program"}
    if enable_instrument_eval:
        by_prog = collections.defaultdict(list)
        for d in scores.get("F1 by program", []):
            for prog, f in d.items():
                by_prog[prog].append(f)
        mean_scores["F1 by program"] = {k: float(np.mean(v)) for k, v in sorted(by_prog.items())}
    for k, v in mean_scores.items():
        print(f"{k}: {v}")
    return mean_scores
