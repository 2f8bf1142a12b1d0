"""End-to-end checks of the MI355X trainer and inference harness on a real GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(variant, dev, **cfg_over):
    from mrmt3.synthetic import T5_SMALL
    cfg = dict(T5_SMALL, **cfg_over)
    if variant == "t5":
        from models.t5 import T5ForConditionalGeneration
        return T5ForConditionalGeneration(cfg).load_golden().to(dev)
    from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
    return T5SegMemV2WithPrev(cfg, 1, 64).load_golden().to(dev)


def test_fused_ce_step_equals_torch_ce_step(dev):
    """Trainer path (fused CE kernel, grads into the flat buffer) == drop-in path (torch CE on the
    returned logits + loss.backward()) on the same batch, dropout off."""
    from mrmt3.synthetic import synth_mel, synth_labels
    from mrmt3.trainer import Trainer
    mel = torch.from_numpy(synth_mel(2)).to(dev)
    lab = torch.from_numpy(synth_labels(2, 256, full=False, seed=5, mean_len=100)).to(dev)
    m1 = _model("t5", dev, dropout_rate=0.0)
    m1.train()
    out = m1(inputs=mel, labels=lab)
    loss1 = torch.nn.functional.cross_entropy(out.view(-1, 1536), lab.view(-1), ignore_index=-100)
    loss1.backward()
    g1 = m1.flat.G.clone()
    m2 = _model("t5", dev, dropout_rate=0.0)
    tr = Trainer(m2, lr=0.0)
    p_before = m2.flat.P.clone()
    loss2 = tr.train_step(mel, lab)
    assert abs(loss1.item() - loss2.item()) < 2e-4
    # dlogits are rounded to bf16 in the fused path, f32->bf16 cast in the other: same values up to rounding
    rel = ((m2.flat.G - g1).norm() / g1.norm()).item()
    assert rel < 2e-2, rel
    assert torch.equal(m2.flat.P, p_before * (1.0 - 0.0))          # lr = 0: AdamW leaves the weights alone


@pytest.mark.parametrize("variant", ["t5", "with_prev"])
def test_trainer_learns_from_audio(dev, variant):
    """Audio -> log-mel -> fwd/bwd -> AdamW for a few steps on one fixed batch: the loss must fall."""
    from mrmt3.synthetic import synth_audio, synth_labels
    from mrmt3.trainer import Trainer
    m = _model(variant, dev)
    tr = Trainer(m, lr=1e-3)
    audio = torch.from_numpy(synth_audio(4)).to(dev)
    lab = torch.from_numpy(synth_labels(4, 256, full=False, seed=9, mean_len=120)).to(dev)
    prev = torch.from_numpy(synth_labels(4, 256, full=False, seed=10, mean_len=120)).to(dev) if variant != "t5" else None
    losses = [tr.train_step(audio, lab, None if prev is None else prev.clone(), audio=True).item() for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.3, losses
    assert int(tr.step_dev.item()) == 8
    ev = tr.eval_loss(audio, lab, None if prev is None else prev.clone(), audio=True).item()
    assert np.isfinite(ev)


def test_weighted_loss_trainer_matches_oracle(dev):
    from mrmt3.synthetic import synth_mel, synth_labels, T5_SMALL, golden_weights
    from mrmt3.trainer import Trainer
    from oracle import t5_ref
    mel = torch.from_numpy(synth_mel(1))
    lab = torch.from_numpy(synth_labels(1, 256, full=False, seed=3, mean_len=150))
    sd = {k: torch.from_numpy(v) for k, v in golden_weights(T5_SMALL).items()}
    with torch.no_grad():
        ref = t5_ref.weighted_ce_loss(t5_ref.forward_logits(sd, T5_SMALL, mel, lab), lab).item()
    m = _model("t5", dev, dropout_rate=0.0)
    tr = Trainer(m, lr=0.0, weighted_loss=True)
    got = tr.eval_loss(mel.to(dev), lab.to(dev)).item()
    assert abs(got - ref) < 2e-3


def test_inference_handler_matches_oracle_pipeline(dev):
    """audio -> InferenceHandler (one log-mel launch for all segments, KV-cached decode) vs the
    oracle's restatement of inference.py (fp32, short max_length)."""
    import inference
    from models.t5 import T5ForConditionalGeneration
    from mrmt3.synthetic import T5_SMALL, golden_weights
    from oracle import logmel_ref, t5_ref
    audio = np.random.RandomState(42).uniform(-1, 1, 40000).astype(np.float32)       # 313 frames -> 2 segments
    model = T5ForConditionalGeneration(T5_SMALL, compute_dtype=torch.float32).load_golden().eval()
    h = inference.InferenceHandler(model=model, device=dev)
    mel_dev, ft = h._preprocess(audio)
    mel_ref, ft_ref, pads = logmel_ref.preprocess(audio)
    assert pads == [256, 57] and np.array_equal(ft, ft_ref)
    np.testing.assert_allclose(mel_dev.cpu().numpy(), mel_ref, atol=1e-4, rtol=0)
    assert (mel_dev[1, 57:] == 0).all()
    results, _ = h.inference(audio, batch_size=8, max_length=24, return_tokens=True)
    sd = {k: torch.from_numpy(v) for k, v in golden_weights(T5_SMALL).items()}
    with torch.no_grad():
        ref_ids = t5_ref.generate_t5(sd, T5_SMALL, torch.from_numpy(mel_ref.astype(np.float32)), max_length=24)
    np.testing.assert_array_equal(results[0], logmel_ref.postprocess_batch(ref_ids.numpy()))
    # audio -> MIDI file end to end (random weights: the notes are meaningless, the plumbing is what is checked)
    import os
    import tempfile
    from oracle import notes_ref
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "out", "song.mid")
        ns = h.inference(audio, outpath=out, batch_size=8, max_length=24)
        assert open(out, "rb").read(4) == b"MThd"
    ref_notes, _, _ = notes_ref.to_event(results, [ft])
    assert [[n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument] for n in ns.notes] == ref_notes
