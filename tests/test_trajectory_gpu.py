"""Trajectory parity on LEARNABLE data, and the pipeline closed end to end.

Every other parity test of the training path is a single step (or a few steps on one fixed batch of random labels).  Here the
tokens are a function of the audio: a segment holds a sine tone (16 pitches x 7 onset slots, 0.2 s long) and its target is what
`mrmt3.tokenizer.Tokenizer` makes of that note (tie section, shifts, program / velocity / pitch tokens: the reference's own target
pipeline, dataset/dataset_2_random.py:108-279).  A T5-small trains on a stream of fresh segments.  (One tone per segment because
that is what a from-scratch T5-small learns to TRANSCRIBE within a test's time: with three tones per segment the loss sits on the
output-prior plateau for more than 3000 steps, with one it leaves the plateau after ~600 steps of 16 segments,
profiles/r06_tone_learning.txt.)

  (a) the bf16 engine with in-kernel dropout 0.1 (the benchmark's arithmetic: bf16 operands, bf16 residual-gradient stream,
      hi/lo attention output, mask generator), the bf16 engine without dropout and the fp32 engine (`precision: 32`,
      config/config.yaml:47 — what the reference trains in) see the SAME segments and must learn alike: smoothed loss curves
      within a stated band of the fp32 one, all three below half the initial loss (tasks/mt3_net.py:27-37 is the step);
  (b) the bf16 + dropout model trains on (1500 steps in all: the loss falls two orders of magnitude below the prior plateau — the
      audio is being read) and then transcribes HELD-OUT segments through the product's own inference path —
      `InferenceHandler.inference` (inference.py:149-234): frames -> log-mel -> greedy decode (hipGraph-replayed steps) ->
      post-processing -> `contrib.note_sequences` -> notes — and `contrib.transcription_metrics` (the mir_eval restatement
      `evaluate.py` scores with) finds the notes: onset F1 above a threshold no untrained or mis-trained model reaches.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PITCHES = list(range(48, 80, 2))            # 16 pitches, 130 Hz .. 830 Hz
SLOTS = [0.10 + 0.25 * i for i in range(7)]  # onset slots (s); a note lasts 0.2 s, so notes never overlap
NOTE_S, SEG_SAMPLES, EVENT_LEN = 0.2, 32768, 64
NOTES_PER_SEGMENT = int(os.environ.get("MRMT3_TRAJ_NOTES", "1"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def tone_segment(rs):
    """(audio [32768] f32, notes): NOTES_PER_SEGMENT tones on distinct slots, 5 ms fades, amplitude 0.3."""
    from contrib.note_sequences import Note
    t = np.arange(SEG_SAMPLES) / 16000.0
    audio = np.zeros(SEG_SAMPLES, np.float32)
    notes = []
    for slot in sorted(rs.choice(len(SLOTS), NOTES_PER_SEGMENT, replace=False)):
        p = int(PITCHES[rs.randint(len(PITCHES))])
        on, off = SLOTS[slot], SLOTS[slot] + NOTE_S
        env = np.clip(np.minimum(t - on, off - t) / 0.005, 0.0, 1.0)
        audio += (0.3 * env * np.sin(2 * np.pi * 440.0 * 2 ** ((p - 69) / 12.0) * (t - on))).astype(np.float32)
        notes.append(Note(on, off, p, 90, 0, False))
    return audio, notes


def tone_batch(rs, n, tk):
    """n fresh segments -> (audio [n, 32768], labels [n, EVENT_LEN] with -100 padding, notes per segment)."""
    from contrib.note_sequences import NoteSequence
    from mrmt3.batching import pad_targets
    audio, rows, all_notes = [], [], []
    for _ in range(n):
        a, notes = tone_segment(rs)
        feats = tk.tokenize(NoteSequence(list(notes), SEG_SAMPLES / 16000.0), SEG_SAMPLES)
        rows.append(tk.row_targets(feats, 0, 256))
        audio.append(a)
        all_notes.append(notes)
    assert max(len(r) for r in rows) < EVENT_LEN
    return np.stack(audio), pad_targets(rows, EVENT_LEN), all_notes


def smooth(x, k=20):
    x = np.asarray(x, np.float64)
    return np.convolve(x, np.ones(k) / k, mode="valid")


def train(dev, dtype, dropout, steps, batch, seed=7, total_steps=1500, lr=3e-4):
    """One model, `steps` optimizer steps over a stream of fresh tone segments (the same stream for a given seed).  Returns the
    model, its trainer, the losses and a `more(n)` that trains n further steps on the same stream."""
    from models.t5 import T5ForConditionalGeneration
    from mrmt3.synthetic import T5_SMALL
    from mrmt3.tokenizer import Tokenizer
    from mrmt3.trainer import Trainer
    from utils import cosine_warmup_lambda
    tk = Tokenizer()
    rs = np.random.RandomState(seed)
    m = T5ForConditionalGeneration(dict(T5_SMALL, dropout_rate=dropout), compute_dtype=dtype).load_golden().to(dev)
    tr = Trainer(m, lr=lr, lr_lambda=cosine_warmup_lambda(20, total_steps, min_lr=1e-4))

    def more(n):
        out = []
        for _ in range(n):
            a, lab, _ = tone_batch(rs, batch, tk)
            out.append(tr.train_step(torch.from_numpy(a).to(dev), lab.to(dev), audio=True))
        torch.cuda.synchronize()
        return [float(x.item()) for x in out]
    return m, tr, more(steps), more


def onset_f1(dev, model, n_segments, seed=1234):
    """Held-out segments through InferenceHandler.inference (one segment = one recording) -> (F1, precision, recall)."""
    from contrib import transcription_metrics as tm
    from contrib.note_sequences import NoteSequence
    from inference import InferenceHandler
    h = InferenceHandler(model=model.eval(), device=dev)
    rs = np.random.RandomState(seed)
    tp = n_ref = n_est = 0
    for _ in range(n_segments):
        audio, notes = tone_segment(rs)
        # (one hop short: `_audio_to_frames` pads a full hop onto aligned audio, inference.py:68, and the model was never shown the
        # one-frame second segment that would make; the last 8 ms of a tone segment are silence)
        est = h.inference(audio[:SEG_SAMPLES - 128], max_length=EVENT_LEN, batch_size=8)
        iv_r, p_r, _ = tm.sequence_to_valued_intervals(NoteSequence(list(notes), SEG_SAMPLES / 16000.0))
        iv_e, p_e, _ = tm.sequence_to_valued_intervals(est)
        n_ref += len(iv_r)
        n_est += len(iv_e)
        if len(iv_e) and len(iv_r):
            tp += len(tm.match_notes(iv_r, tm.midi_to_hz(p_r), iv_e, tm.midi_to_hz(p_e), onset_tolerance=0.05, offset_ratio=None))
    prec, rec = tp / max(n_est, 1), tp / max(n_ref, 1)
    return tm.f_measure(prec, rec), prec, rec


def test_bf16_with_dropout_trains_like_fp32_on_learnable_data_and_transcribes_held_out_audio(dev):
    steps = int(os.environ.get("MRMT3_TRAJ_STEPS", "300"))
    B = 16
    runs = {}
    for name, dtype, p in (("fp32", torch.float32, 0.0), ("bf16", torch.bfloat16, 0.0), ("bf16+dropout", torch.bfloat16, 0.1)):
        m, tr, losses, more = train(dev, dtype, p, steps, B)
        assert all(np.isfinite(losses)), name
        runs[name] = (m, tr, losses, more)
        print("%-13s loss %.3f -> %s" % (name, losses[0], " ".join("%.3f" % v for v in smooth(losses)[::max(1, steps // 10)])))
    def windows(x, w=50):                                  # medians over windows of 50 steps: robust to the single-batch spikes
        x = np.asarray(x, np.float64)                      # every Adam run at this learning rate has (at its own steps)
        return np.array([np.median(x[i:i + w]) for i in range(0, len(x) - w + 1, w)])
    first = runs["fp32"][2][0]
    ref = windows(runs["fp32"][2])
    assert ref[-1] < 0.5 * first, (first, ref[-1])                       # the data IS learnable in this many steps
    # The yardstick for "the same trajectory": training is a chaotic map — once the loss sits on the output-prior plateau, ANY
    # perturbation moves the single-batch spikes around.  So the fp32 engine is run once more with its learning rate scaled by
    # 1 + 1e-6 (one part in a million: far below any arithmetic difference between the engines), and the gap between the two
    # fp32 runs is what "numerically the same" looks like on this data (measured: 0 / 0 / 0 / 1 % / 20 % / 10 % over the six
    # windows); the bf16 engine may be at most 3 x that far from fp32, and within 2 % during the plunge, before chaos has had
    # time to act (measured 0.04 % / 0.09 % / 1.3 % / 9 % / 3 % / 3 %).  Dropout makes the plunge itself slower — the masked
    # model is a weaker one (measured 1.35 x / 1.55 x fp32's loss in the first two windows) — and then joins the others: from
    # step 100 on the same 3 x band (measured 3 % / 0.7 % / 9 % / 8 %).
    twin = windows(train(dev, torch.float32, 0.0, steps, B, lr=3e-4 * (1 + 1e-6))[2])
    self_gap = np.abs(twin - ref) / ref
    print("fp32          window medians %s" % " ".join("%.4f" % v for v in ref))
    print("fp32 twin     window medians %s; gap to fp32 %s" % (" ".join("%.4f" % v for v in twin), " ".join("%.4f" % v for v in self_gap)))
    for name in ("bf16", "bf16+dropout"):
        s = windows(runs[name][2])
        assert s[-1] < 0.5 * first, (name, first, s[-1])
        rel = np.abs(s - ref) / ref
        print("%-13s window medians %s; gap to fp32 %s" % (name, " ".join("%.4f" % v for v in s), " ".join("%.4f" % v for v in rel)))
        if name == "bf16":
            assert rel[0] < 0.02, (name, rel)
            assert rel.max() < max(3.0 * self_gap.max(), 0.05), (name, rel, self_gap)
        else:
            assert np.all(s[:2] >= 0.98 * ref[:2]) and np.all(s[:2] < 2.0 * ref[:2]), (name, s, ref)
            assert rel[2:].max() < max(3.0 * self_gap.max(), 0.15), (name, rel, self_gap)
    # same weights both ways at the end?  not bit for bit — but the three models are the same FUNCTION: evaluation loss (no dropout)
    from mrmt3.tokenizer import Tokenizer
    a, lab, _ = tone_batch(np.random.RandomState(99), 32, Tokenizer())
    ev = {n: float(runs[n][1].eval_loss(torch.from_numpy(a).to(dev), lab.to(dev), audio=True).item()) for n in runs}
    print("held-out evaluation loss:", ev)
    assert abs(ev["bf16"] - ev["fp32"]) < 0.05 * ev["fp32"] + 0.02, ev
    assert ev["bf16+dropout"] < 1.25 * ev["fp32"] + 0.05, ev
    # (b) the bf16 + dropout model trains on until it READS the audio (the loss leaves the output-prior plateau), then
    # audio -> InferenceHandler -> notes -> onset F1 on held-out tones
    total = int(os.environ.get("MRMT3_TRAJ_TOTAL", "1500"))
    m, tr, losses, more = runs["bf16+dropout"]
    plateau = float(np.median(losses[-100:]))
    tail = more(total - steps)
    late = float(np.median(tail[-100:]))
    print("bf16+dropout  prior plateau %.4f at step %d -> %.5f at step %d" % (plateau, steps, late, total))
    assert late < 0.1 * plateau and late < 0.02, (plateau, late)
    f1, prec, rec = onset_f1(dev, m, 24)
    print("held-out onset F1 %.3f (precision %.3f, recall %.3f) after %d steps of %d segments" % (f1, prec, rec, total, B))
    # measured: held-out evaluation loss fp32 0.114 / bf16 0.118 / bf16 + dropout 0.142 after 300 steps; training loss 0.154 ->
    # 0.00098 at step 1500; held-out onset F1 0.958 (23 of 24 notes found, none invented)
    floor = float(os.environ.get("MRMT3_TRAJ_F1", "0.6"))
    assert f1 >= floor, (f1, prec, rec)
    for _, tr_, _, _ in runs.values():
        tr_.close()
