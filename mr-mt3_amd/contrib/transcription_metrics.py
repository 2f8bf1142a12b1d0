"""Note-level transcription scores — the arithmetic evaluate.py takes from `mir_eval.transcription`
(precision_recall_f1_overlap, offset_ratio=None or 0.2) and `mir_eval.util.f_measure`, restated from
mir_eval's published algorithm (mir_eval is a third-party dependency of the reference, absent from
/root/reference and from this image; `requirements.txt` does not pin a version):

  * a reference and an estimated note match when their onsets are within 50 ms (distances rounded to 6
    decimals, `<=`), their pitches within 50 cents (1200 * |log2 f_ref - log2 f_est|), and — unless
    `offset_ratio` is None — their offsets within max(offset_ratio * ref_duration, 50 ms);
  * the score uses a MAXIMUM-cardinality one-to-one matching of that bipartite graph;
    precision = matched / n_est, recall = matched / n_ref, F = 2PR / (P + R);
  * average overlap ratio = mean over matched pairs of intersection / union of the two intervals.
P, R and F depend only on the matching's size, which is unique; the overlap ratio may differ from
mir_eval's in the last digits when several maximum matchings exist.
"""
from __future__ import annotations

import numpy as np

N_DECIMALS = 6


def f_measure(precision: float, recall: float, beta: float = 1.0) -> float:
    if precision == 0 and recall == 0:
        return 0.0
    return (1 + beta ** 2) * precision * recall / ((beta ** 2) * precision + recall)


def midi_to_hz(notes):
    """librosa.midi_to_hz: 440 * 2 ** ((m - 69) / 12)."""
    return 440.0 * (2.0 ** ((np.asanyarray(notes, dtype=np.float64) - 69.0) / 12.0))


def _validate(ref_intervals, ref_pitches, est_intervals, est_pitches):
    for iv, name in ((ref_intervals, "reference"), (est_intervals, "estimated")):
        if iv.ndim != 2 or iv.shape[1] != 2:
            raise ValueError(f"{name} intervals should be n-by-2, got shape {iv.shape}")
        if (iv < 0).any():
            raise ValueError(f"negative {name} interval time")
        if (iv[:, 1] <= iv[:, 0]).any():
            raise ValueError(f"all {name} interval durations must be strictly positive")
    if ref_intervals.shape[0] != ref_pitches.shape[0] or est_intervals.shape[0] != est_pitches.shape[0]:
        raise ValueError("intervals and pitches have different lengths")
    if ref_pitches.size and ref_pitches.min() <= 0 or est_pitches.size and est_pitches.min() <= 0:
        raise ValueError("pitches must be positive")


def _max_bipartite_matching(adj, n_left):
    """Maximum-cardinality matching by augmenting paths (iterative DFS).  adj[u] = right vertices adjacent to
    left vertex u.  Returns {left: right}.  The graphs here are sparse (a note has a handful of candidates)."""
    match_l, match_r = {}, {}
    for root in range(n_left):
        if not adj.get(root):
            continue
        parent, seen, end = {}, set(), None
        stack = [(root, iter(adj[root]))]
        while stack and end is None:
            left, it = stack[-1]
            for v in it:
                if v in seen:
                    continue
                seen.add(v)
                parent[v] = left
                owner = match_r.get(v)
                if owner is None:
                    end = v
                else:
                    stack.append((owner, iter(adj.get(owner, ()))))
                break
            else:
                stack.pop()
        while end is not None:                  # flip the path root -> ... -> end
            left = parent[end]
            prev = match_l.get(left)
            match_l[left], match_r[end] = end, left
            end = None if left == root else prev
    return match_l


def match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
                offset_ratio=0.2, offset_min_tolerance=0.05, strict=False):
    """Sorted list of (ref_index, est_index) of a maximum matching."""
    cmp = np.less if strict else np.less_equal
    onset = np.around(np.abs(np.subtract.outer(ref_intervals[:, 0], est_intervals[:, 0])), decimals=N_DECIMALS)
    hit = cmp(onset, onset_tolerance)
    cents = np.abs(1200.0 * np.subtract.outer(np.log2(ref_pitches), np.log2(est_pitches)))
    hit &= cmp(cents, pitch_tolerance)
    if offset_ratio is not None:
        off = np.around(np.abs(np.subtract.outer(ref_intervals[:, 1], est_intervals[:, 1])), decimals=N_DECIMALS)
        tol = np.maximum(offset_ratio * (ref_intervals[:, 1] - ref_intervals[:, 0]), offset_min_tolerance)
        hit &= cmp(off, tol.reshape(-1, 1))
    adj = {}
    for r, e in zip(*np.where(hit)):
        adj.setdefault(int(e), []).append(int(r))          # estimated notes on the left, like mir_eval's graph
    m = _max_bipartite_matching(adj, est_intervals.shape[0])
    return sorted((r, e) for e, r in m.items())


def average_overlap_ratio(ref_intervals, est_intervals, matching) -> float:
    ratios = []
    for r, e in matching:
        a, b = ref_intervals[r], est_intervals[e]
        ratios.append((min(a[1], b[1]) - max(a[0], b[0])) / (max(a[1], b[1]) - min(a[0], b[0])))
    return float(np.mean(ratios)) if ratios else 0


def precision_recall_f1_overlap(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05,
                                pitch_tolerance=50.0, offset_ratio=0.2, offset_min_tolerance=0.05, strict=False,
                                beta=1.0):
    ref_intervals = np.asarray(ref_intervals, dtype=np.float64).reshape(-1, 2)
    est_intervals = np.asarray(est_intervals, dtype=np.float64).reshape(-1, 2)
    ref_pitches = np.asarray(ref_pitches, dtype=np.float64)
    est_pitches = np.asarray(est_pitches, dtype=np.float64)
    _validate(ref_intervals, ref_pitches, est_intervals, est_pitches)
    if len(ref_pitches) == 0 or len(est_pitches) == 0:
        return 0.0, 0.0, 0.0, 0.0
    matching = match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance, pitch_tolerance,
                           offset_ratio, offset_min_tolerance, strict)
    precision = len(matching) / len(est_pitches)
    recall = len(matching) / len(ref_pitches)
    return precision, recall, f_measure(precision, recall, beta), average_overlap_ratio(ref_intervals, est_intervals, matching)


def sequence_to_valued_intervals(ns, min_midi_pitch=21, max_midi_pitch=108):
    """note_seq.sequences_lib.sequence_to_valued_intervals: (intervals [n,2], MIDI pitches, velocities) of
    the notes inside the piano range with a non-zero duration."""
    rows = [(n.start_time, n.end_time, n.pitch, n.velocity) for n in ns.notes
            if min_midi_pitch <= n.pitch <= max_midi_pitch and n.end_time != n.start_time]
    a = np.array(rows, dtype=np.float64).reshape(-1, 4)
    return a[:, :2], a[:, 2], a[:, 3]
