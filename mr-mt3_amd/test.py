"""MI355X evaluation driver — the build's counterpart of the reference's test.py:15-140: transcribe every
recording of `cfg.eval.audio_dir` to a MIDI file with `inference.InferenceHandler`, then score the directory
with `evaluate.evaluate_main`.

    python test.py --config-dir /path/to/reference/config --config-name config_slakh_segmem \\
        model=MT3NetSegMemV2WithPrev path=/ckpt/last.ckpt eval.audio_dir='/data/slakh/test/*/mix_16k.wav' \\
        eval.exp_tag_name=run1 +output_dir=outputs

Same config keys as the reference (`path`, `eval.{audio_dir, midi_dir, eval_dataset, exp_tag_name, batch_size,
contiguous_inference, eval_first_n_examples, load_weights_strict}`, `dataset.test.root_dir`); `.ckpt` files go through
`load_from_checkpoint`, `.pt/.pth` through `load_state_dict(strict=False)`; `mel_norm` is off only for the official
`pretrained/mt3.pth` (test.py:123).  Differences: WAV files are read without librosa (`contrib.audio_io`), and with a
segment-memory model the recordings are decoded several at a time in lockstep (`InferenceHandler.inference_many`,
`+eval.songs_per_batch=8`), which produces the same tokens as one at a time.
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from contrib import audio_io  # noqa: E402
from evaluate import evaluate_main  # noqa: E402
from inference import InferenceHandler  # noqa: E402
from mrmt3 import hydra_lite  # noqa: E402


def _outpath(fname, eval_dataset, out_root):
    if eval_dataset == "Slakh":
        return os.path.join(out_root, fname.split("/")[-2], "mix.mid")
    if eval_dataset in ("ComMU", "NSynth"):
        return os.path.join(out_root, fname.split("/")[-1].replace(".wav", ".mid"))
    raise ValueError("Invalid dataset name.")


def get_scores(model, eval_audio_dir=None, mel_norm=True, eval_dataset="Slakh", exp_tag_name="test_midis",
               ground_truth_midi_dir=None, verbose=True, contiguous_inference=False, use_tf_spectral_ops=False,
               batch_size=8, max_length=1024, output_dir=".", songs_per_batch=8):
    """test.py:15-81.  Returns the mean scores of `evaluate_main`."""
    handler = InferenceHandler(model=model, device=torch.device("cuda"), mel_norm=mel_norm,
                               contiguous_inference=contiguous_inference, use_tf_spectral_ops=use_tf_spectral_ops)

    def read(fname):
        audio, _ = audio_io.load(fname, sr=16000)
        if eval_dataset == "NSynth":                            # test.py:38-39
            audio = np.pad(audio, (int(0.05 * 16000), 0), "constant", constant_values=0)
        return audio

    files = list(eval_audio_dir)
    out_root = os.path.join(output_dir, exp_tag_name)
    if verbose:
        print("Total songs:", len(files))
    lockstep = hasattr(model, "generate_songs") and songs_per_batch > 1
    if lockstep:
        for i in range(0, len(files), songs_per_batch):
            group = files[i:i + songs_per_batch]
            handler.inference_many([read(f) for f in group], outpaths=[_outpath(f, eval_dataset, out_root) for f in group],
                                   max_length=max_length)
    else:
        for fname in files:
            handler.inference(audio=read(fname), audio_path=fname, outpath=_outpath(fname, eval_dataset, out_root),
                              batch_size=batch_size, max_length=max_length, verbose=verbose)
    if verbose:
        print("Evaluating...")
    scores = evaluate_main(dataset_name=eval_dataset, test_midi_dir=out_root, ground_truth_midi_dir=ground_truth_midi_dir)
    if verbose:
        for key in sorted(scores):
            if not isinstance(scores[key], dict):
                print("{}: {:.4}".format(key, scores[key]))
    return scores


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", "--config-path", dest="config_dir", required=True)   # both spellings appear in the reference's scripts
    ap.add_argument("--config-name", default="config")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args(argv)
    cfg = hydra_lite.compose(a.config_dir, a.config_name, a.overrides)
    path = str(cfg.get("path") or "")
    assert path, "path=<weights> is required"                                               # test.py:85
    assert path.endswith((".pt", ".pth", ".ckpt")), "Only .pt, .pth, .ckpt files are supported."
    assert cfg.eval.get("exp_tag_name") and cfg.eval.get("audio_dir")
    task = hydra_lite.instantiate(cfg.model, optim_cfg=cfg.optim)
    print(f"Loading weights from: {path}")
    if path.endswith(".ckpt"):
        cls = type(task)
        task = cls.load_from_checkpoint(path, config=cfg.model.config, optim_cfg=cfg.optim)
        model = task.model
    else:
        model = task.model
        from mrmt3.checkpoint import read_checkpoint
        strict = cfg.eval.get("load_weights_strict")
        model.load_state_dict(read_checkpoint(path)["state_dict"], strict=bool(strict) if strict is not None else False)
    model.eval()
    files = sorted(glob.glob(str(cfg.eval.audio_dir)))
    if cfg.eval.eval_dataset == "NSynth":
        files = [d for d in files if "vocal" not in d and "mallet" not in d]                # test.py:116-118
    if cfg.eval.get("eval_first_n_examples"):
        files = files[:int(cfg.eval.eval_first_n_examples)]
    mel_norm = "pretrained/mt3.pth" not in path
    gt = cfg.eval.get("midi_dir") or cfg.dataset.test.root_dir
    return get_scores(model, eval_audio_dir=files, mel_norm=mel_norm, eval_dataset=str(cfg.eval.eval_dataset),
                      exp_tag_name=str(cfg.eval.exp_tag_name), ground_truth_midi_dir=str(gt),
                      contiguous_inference=bool(cfg.eval.get("contiguous_inference", False)),
                      use_tf_spectral_ops=bool(cfg.eval.get("use_tf_spectral_ops", False)),
                      batch_size=int(cfg.eval.get("batch_size", 8)), output_dir=str(cfg.get("output_dir", ".")),
                      songs_per_batch=int(cfg.eval.get("songs_per_batch", 8)))


if __name__ == "__main__":
    main()
