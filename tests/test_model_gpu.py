"""Model-level parity on a real MI355X against (a) the golden vectors recorded from the reference
itself and (b) the CPU oracle on identical seeded inputs.

Tolerances (stated per test):
  fp32 compute path : logits within 2e-4 of the reference, loss within 2e-5, greedy token ids bit-exact
  bf16 compute path : loss within 1e-3 (north_star).  Logits: north_star's 1e-3 is not reachable by ANY bf16-operand
                      arithmetic on this model — the reference itself, run under torch.autocast(bfloat16), deviates from
                      its own fp32 logits by max|d| 3.6e-2..4.8e-2, rel-L2 7.3e-3..8.0e-3 (tests/golden/bf16_bound.npz,
                      recorded by make_golden.py --bf16-bound).  The HIP path is held to THAT, per variant: max|d| and
                      rel-L2 no worse than the reference's autocast run, absolute cap 4e-2 (measured 2.3e-2..2.9e-2;
                      lm_head in exact f32 changes nothing, profiles/tools/bf16_logit_gap.py: the deviation is the
                      accumulated operand rounding of 16 layers)
  bf16 gradients    : per tensor, rel-L2 vs the fp32 oracle gradient <= 1.5 x what the reference's autocast gradient
                      loses on that tensor (+2e-3), < 3e-2 absolute, cosine > 0.9995
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
VARIANTS = ["t5", "segmem_v1", "segmem_v2", "segmem_v2_with_prev"]


def _build(variant, dtype, dev):
    from mrmt3.synthetic import T5_SMALL
    if variant == "t5":
        from models.t5 import T5ForConditionalGeneration as M
        m = M(T5_SMALL, compute_dtype=dtype)
    else:
        import importlib
        mod, cls = {"segmem_v1": ("models.t5_segmem", "T5SegMem"), "segmem_v2": ("models.t5_segmem_v2", "T5SegMemV2"),
                    "segmem_v2_with_prev": ("models.t5_segmem_v2_with_prev", "T5SegMemV2WithPrev")}[variant]
        m = getattr(importlib.import_module(mod), cls)(T5_SMALL, segmem_num_layers=1, segmem_length=64,
                                                       compute_dtype=dtype)
    return m.load_golden().to(dev).eval()


def _inputs(dev):
    from mrmt3.synthetic import synth_mel, synth_labels
    B = 2
    return (torch.from_numpy(synth_mel(B)).to(dev), torch.from_numpy(synth_labels(B, full=True)).to(dev),
            torch.from_numpy(synth_labels(B, full=False, seed=777)).to(dev),
            torch.from_numpy(synth_labels(B, full=False, seed=999)).to(dev))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("tag", ["full", "pad"])
def test_fp32_logits_match_reference(dev, golden, variant, tag):
    m = _build(variant, torch.float32, dev)
    mel, lab_full, lab_pad, prev = _inputs(dev)
    lab = lab_full if tag == "full" else lab_pad
    with torch.no_grad():
        logits = m(inputs=mel, labels=lab, targets_prev=prev.clone())
    got = logits.reshape(-1)[torch.from_numpy(golden[f"{variant}.{tag}.logit_idx"]).to(dev)].cpu().numpy()
    np.testing.assert_allclose(got, golden[f"{variant}.{tag}.logit_val"], atol=2e-4, rtol=0)
    loss = torch.nn.functional.cross_entropy(logits.view(-1, 1536).double(), lab.view(-1), ignore_index=-100).item()
    assert abs(loss - float(golden[f"{variant}.{tag}.loss"])) < 2e-5
    assert (logits.argmax(-1).cpu().numpy() == golden[f"{variant}.{tag}.argmax"]).mean() > 0.999


@pytest.mark.parametrize("variant", VARIANTS)
def test_bf16_logits_and_loss(dev, golden, variant):
    m = _build(variant, torch.bfloat16, dev)
    mel, lab_full, lab_pad, prev = _inputs(dev)
    with torch.no_grad():
        logits = m(inputs=mel, labels=lab_pad, targets_prev=prev.clone())
    idx = torch.from_numpy(golden[f"{variant}.pad.logit_idx"]).to(dev)
    got = logits.reshape(-1)[idx].cpu().numpy()
    ref = golden[f"{variant}.pad.logit_val"]
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    loss = torch.nn.functional.cross_entropy(logits.view(-1, 1536).double(), lab_pad.view(-1), ignore_index=-100).item()
    print(variant, "bf16: rel-L2 %.3e max|d| %.3e dloss %.2e" % (rel, np.abs(got - ref).max(), loss - float(golden[f"{variant}.pad.loss"])))
    assert abs(loss - float(golden[f"{variant}.pad.loss"])) < 1e-3       # north_star tolerance on the loss
    # logits: no worse than what the reference's own bf16-autocast run loses on the same inputs (see module docstring)
    bound = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_bound.npz"))
    assert rel <= float(bound[f"{variant}.pad.autocast_rel_l2"])
    max_d = float(np.abs(got - ref).max())
    assert max_d <= float(bound[f"{variant}.pad.autocast_max_abs"]) and max_d < 4e-2
    # the sampled logits of the autocast run: the HIP path is also closer to fp32 than autocast is, sample by sample on average
    assert np.abs(got - ref).mean() <= np.abs(bound[f"{variant}.pad.autocast_logit_val"] - ref).mean()
    assert (logits.argmax(-1).cpu().numpy() == golden[f"{variant}.pad.argmax"]).mean() > 0.97    # near-tied logits flip under bf16


def test_state_dict_schema_roundtrip(dev):
    """193 tensors + 2 aliases + inv_freq buffers under the reference's keys; strict=False load."""
    from mrmt3.synthetic import T5_SMALL, state_dict_shapes
    m = _build("segmem_v2_with_prev", torch.bfloat16, dev)
    sd = m.state_dict()
    for k, shp in state_dict_shapes(T5_SMALL, 1).items():
        assert tuple(sd[k].shape) == tuple(shp), k
    for alias, src in (("encoder.embed_tokens.weight", "proj.weight"), ("decoder.embed_tokens.weight", "decoder_embed_tokens.weight"),
                       ("segmem_encoder.embed_tokens.weight", "segmem_proj.weight")):
        assert sd[alias].data_ptr() == sd[src].data_ptr()
    assert sd["encoder.pos_emb.inv_freq"].shape == (256,)
    m2 = _build("segmem_v2_with_prev", torch.bfloat16, dev)
    m2.reset_parameters(seed=5)
    missing, unexpected = m2.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected
    assert torch.equal(m2.flat.P, m.flat.P)
    # q|k|v adjacency survives the device move: fused view == concatenation
    q, k, v = (sd[f"decoder.block.3.layer.0.SelfAttention.{n}.weight"] for n in "qkv")
    assert torch.equal(m.flat.W("decoder.3.qkv", torch.float32), torch.cat([q, k, v], 0))


@pytest.mark.parametrize("variant", ["t5", "segmem_v2_with_prev", "segmem_v1", "segmem_v2"])
def test_bf16_gradients_vs_oracle_autograd(dev, variant):
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
    from oracle import t5_ref
    torch.set_num_threads(8)
    B = 2
    mel = torch.from_numpy(synth_mel(B))
    lab = torch.from_numpy(synth_labels(B, 256, full=False, seed=777, mean_len=120))
    prev = torch.from_numpy(synth_labels(B, 256, full=False, seed=999, mean_len=120))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 0 if variant == "t5" else 1).items()}
    logits = t5_ref.forward_logits(sd, T5_SMALL, mel, lab, variant=variant, targets_prev=prev.clone())
    ref_loss = t5_ref.ce_loss(logits, lab)
    ref_loss.backward()
    m = _build(variant, torch.bfloat16, dev)
    out = m(inputs=mel.to(dev), labels=lab.to(dev), targets_prev=prev.clone().to(dev))
    loss = torch.nn.functional.cross_entropy(out.view(-1, 1536), lab.to(dev).view(-1), ignore_index=-100)
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-3
    bound = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_bound.npz"))
    ref_rel = dict(zip(bound[f"{variant}.grad_names"].tolist(), bound[f"{variant}.grad_rel_l2"].tolist()))
    worst = (1.0, 0.0, "")
    for k, ref in sd.items():
        g = m.flat.grad(k).cpu()
        r = ref.grad
        if r is None or r.norm() == 0:
            assert g.norm() < 1e-6, k
            continue
        cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        rel = ((g - r).norm() / r.norm()).item()
        if cos < worst[0]:
            worst = (cos, rel, k)
        assert cos > 0.9995 and rel < 3e-2, (k, cos, rel)
        if k in ref_rel:        # what the reference's own bf16-autocast gradient loses on this tensor
            assert rel <= 1.5 * ref_rel[k] + 2e-3, (k, rel, ref_rel[k])
    print(variant, "worst grad tensor:", worst)
    # every parameter exposes its slice of the flat buffer as .grad
    p = dict(m.named_parameters())["lm_head.weight"]
    assert p.grad is not None and p.grad.data_ptr() == m.flat.grad("lm_head.weight").data_ptr()


@pytest.mark.parametrize("variant", VARIANTS)
def test_fp32_gradients_match_oracle(dev, variant):
    """The hand-written backward at the reference's own training precision (`precision: 32`,
    config/config_slakh_segmem.yaml:47): an fp32 engine — exact-f32 MFMA products, f32 attention backward, f32 row
    kernels — against oracle/t5_ref.py autograd on the same inputs.  Tolerance: every gradient tensor within
    rel-L2 1e-4 (measured ~1e-6), loss within 2e-5.  This is the tight pin of the tape in mrmt3/engine.py; the bf16
    tests above can only say "within bf16 noise"."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
    from oracle import t5_ref
    torch.set_num_threads(8)
    B = 2
    mel = torch.from_numpy(synth_mel(B))
    lab = torch.from_numpy(synth_labels(B, 256, full=False, seed=777, mean_len=120))
    prev = torch.from_numpy(synth_labels(B, 256, full=False, seed=999, mean_len=120))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 0 if variant == "t5" else 1).items()}
    logits = t5_ref.forward_logits(sd, T5_SMALL, mel, lab, variant=variant, targets_prev=prev.clone())
    ref_loss = t5_ref.ce_loss(logits, lab)
    ref_loss.backward()
    m = _build(variant, torch.float32, dev)
    out = m(inputs=mel.to(dev), labels=lab.to(dev), targets_prev=prev.clone().to(dev))
    loss = torch.nn.functional.cross_entropy(out.view(-1, 1536), lab.to(dev).view(-1), ignore_index=-100)
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    worst = (0.0, "")
    for k, ref in sd.items():
        g = m.flat.grad(k).cpu()
        r = ref.grad
        if r is None or r.norm() == 0:
            assert g.norm() < 1e-7, k
            continue
        rel = ((g - r).norm() / r.norm()).item()
        worst = max(worst, (rel, k))
        assert rel < 1e-4, (k, rel)
    print(variant, "fp32 gradients: worst rel-L2 %.3e (%s)" % worst)


def test_fp32_trainer_step_matches_torch_adamw_on_the_oracle(dev):
    """Three optimizer steps of the fp32 engine through mrmt3.trainer.Trainer (dropout off) against torch.optim.AdamW
    driving the oracle: weights within 1e-4 of a 3e-3 move after the steps (the fp32 path is a training path, not only a gradient check)."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
    from mrmt3.trainer import Trainer
    from models.t5 import T5ForConditionalGeneration
    from oracle import t5_ref
    torch.set_num_threads(8)
    cfg = dict(T5_SMALL, dropout_rate=0.0)
    mel = torch.from_numpy(synth_mel(2, seed=5))
    lab = torch.from_numpy(synth_labels(2, 128, full=False, seed=6, mean_len=80))
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in golden_weights(cfg).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3)
    ref_losses = []
    for _ in range(3):
        opt.zero_grad()
        l = t5_ref.ce_loss(t5_ref.forward_logits(sd, cfg, mel, lab, variant="t5"), lab)
        l.backward()
        opt.step()
        ref_losses.append(l.item())
    m = T5ForConditionalGeneration(cfg, compute_dtype=torch.float32).load_golden().to(dev)
    tr = Trainer(m, lr=1e-3, graph=False)
    losses = [tr.train_step(mel.to(dev), lab.to(dev)).item() for _ in range(3)]
    assert np.allclose(losses, ref_losses, atol=5e-5), (losses, ref_losses)
    # AdamW's first steps move every weight by ~lr * sign(g): an element whose gradient is near zero can flip sign on a
    # 1e-7 difference and land 2e-3 away, so single elements are not the measure — the update as a whole is
    w0 = golden_weights(cfg)
    worst = 0.0
    for k, ref in sd.items():
        p0 = torch.from_numpy(w0[k])
        du, dr = m.flat.master(k).cpu() - p0, ref.detach() - p0
        rel = ((du - dr).norm() / dr.norm()).item()
        worst = max(worst, rel)
        assert rel < 2e-2, (k, rel)
        assert ((du - dr).abs() > 1e-4).float().mean().item() < 1e-3, k        # the sign-flipped few
    print("fp32 trainer: worst rel-L2 of the three-step update vs torch.optim.AdamW on the oracle: %.2e" % worst)


def test_fp32_training_with_dropout_draws_the_bf16_masks_and_replays_from_graphs(dev):
    """The fp32 engine with dropout ON: (i) it draws the masks of the bf16 engine (same seed, site ids and step counter), so
    its loss trajectory stays within bf16 noise of the bf16 engine's — a different mask anywhere moves the loss by ~5e-2;
    (ii) the step replayed from hipGraphs equals the eager step bit for bit, like the bf16 path."""
    from mrmt3.synthetic import T5_SMALL, synth_mel, synth_labels
    from mrmt3.trainer import Trainer
    from models.t5 import T5ForConditionalGeneration
    mel = torch.from_numpy(synth_mel(2, seed=8)).to(dev)
    lab = torch.from_numpy(synth_labels(2, 128, full=False, seed=9, mean_len=80)).to(dev)
    runs = {}
    for name, dtype, graph in (("bf16", torch.bfloat16, False), ("f32", torch.float32, False), ("f32_graph", torch.float32, True)):
        m = T5ForConditionalGeneration(T5_SMALL, compute_dtype=dtype).load_golden().to(dev)
        tr = Trainer(m, lr=1e-3, graph=graph)
        losses = [tr.train_step(mel if dtype == torch.float32 else mel.bfloat16(), lab).item() for _ in range(5)]
        torch.cuda.synchronize()
        assert tr.graph_captured == graph
        runs[name] = (losses, m.flat.P.clone())
    assert all(np.isfinite(runs[k][0]).all() for k in runs)
    # same masks: the first steps agree to bf16 noise (measured 6e-4 .. 3e-3; another mask anywhere moves a loss by ~5e-2);
    # later steps drift apart as the bf16 rounding of three updates at lr 1e-3 accumulates (1.3e-2 at step 4)
    d = np.abs(np.array(runs["f32"][0]) - np.array(runs["bf16"][0]))
    assert d[:3].max() < 5e-3 and d.max() < 3e-2, (runs["f32"][0], runs["bf16"][0])
    assert runs["f32"][0][-1] < runs["f32"][0][1]                       # it trains (step 0 is before the first update)
    assert np.allclose(runs["f32"][0], runs["f32_graph"][0], rtol=0, atol=2e-6)
    assert torch.equal(runs["f32"][1], runs["f32_graph"][1])


def test_lightning_style_steps_of_the_segment_memory_tasks(dev):
    """MT3NetSegMemV2WithPrev (3-tuple batches, cosine schedule) and its FineTune subclass (bare AdamW) driven the way
    Lightning drives them: training_step -> backward -> optimizer step, validation_step under no_grad."""
    from mrmt3.synthetic import T5_SMALL
    from tasks.mt3_net_segmem_v2_with_prev import MT3NetSegMemV2WithPrev
    from tasks.mt3_net_segmem_v2_with_prev_finetune import MT3NetSegMemV2WithPrevFineTune
    optim_cfg = dict(lr=1e-3, warmup_steps=1, num_steps_per_epoch=10, num_epochs=1, min_lr=1e-4)
    cfg = dict(T5_SMALL, dropout_rate=0.0, segmem_num_layers=1, segmem_length=64)
    mel, _, lab_pad, prev = _inputs(dev)
    lab, prv = lab_pad[:, :128].contiguous(), prev[:, :128].contiguous()
    for cls in (MT3NetSegMemV2WithPrev, MT3NetSegMemV2WithPrevFineTune):
        task = cls(dict(cfg), optim_cfg)
        task.model.load_golden()
        task.to(dev).train()
        conf = task.configure_optimizers()
        if isinstance(conf, tuple):
            (opt,), (sched,) = conf
        else:
            opt, sched = conf, None                                 # finetune: bare AdamW (…_finetune.py:11-20)
        assert isinstance(opt, torch.optim.AdamW)
        losses = []
        for it in range(5):
            opt.zero_grad()
            loss = task.training_step((mel, lab, prv.clone()), it)
            loss.backward()
            opt.step()
            if sched is not None:
                sched["scheduler"].step()
            losses.append(loss.item())
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        task.eval()
        task.validation_step((mel, lab, prv.clone()), 0)
        assert "val_loss" in task.logged and np.isfinite(float(task.logged["val_loss"]))


def test_two_tuple_task_classes_step_and_weighted_loss_value(dev):
    """MT3NetSegMem (V1), MT3NetSegMemV2 and MT3NetWeightedLoss: (inputs, targets) batches; the weighted task's loss
    equals the oracle's weighted cross-entropy of the same logits."""
    from mrmt3.synthetic import T5_SMALL
    from oracle import t5_ref
    from tasks.mt3_net import MT3NetWeightedLoss
    from tasks.mt3_net_segmem import MT3NetSegMem
    from tasks.mt3_net_segmem_v2 import MT3NetSegMemV2
    optim_cfg = dict(lr=1e-3, warmup_steps=1, num_steps_per_epoch=10, num_epochs=1, min_lr=1e-4)
    mel, _, lab_pad, _ = _inputs(dev)
    lab = lab_pad[:, :128].contiguous()
    for cls, extra in ((MT3NetSegMem, dict(segmem_num_layers=1, segmem_length=64)),
                       (MT3NetSegMemV2, dict(segmem_num_layers=1, segmem_length=64)), (MT3NetWeightedLoss, {})):
        task = cls(dict(T5_SMALL, dropout_rate=0.0, **extra), optim_cfg)
        task.model.load_golden()
        task.to(dev).train()
        (opt,), _ = task.configure_optimizers()
        opt.zero_grad()
        loss = task.training_step((mel, lab), 0)
        loss.backward()
        opt.step()
        assert np.isfinite(loss.item()) and task.model.flat.G.abs().sum().item() > 0
        if cls is MT3NetWeightedLoss:
            task.model.load_golden()
            task.eval()
            with torch.no_grad():
                logits = task.model(inputs=mel, labels=lab)
                got = task.training_step((mel, lab), 1).item()
            want = t5_ref.weighted_ce_loss(logits.float().cpu(), lab.cpu()).item()
            assert abs(got - want) < 2e-4, (got, want)


def test_grad_accumulation_and_zero_grad(dev):
    m = _build("t5", torch.bfloat16, dev)
    mel, lab_full, lab_pad, _ = _inputs(dev)
    lab = lab_pad[:, :128].contiguous()

    def step():
        out = m(inputs=mel, labels=lab)
        torch.nn.functional.cross_entropy(out.view(-1, 1536), lab.view(-1), ignore_index=-100).backward()
    step()
    g1 = m.flat.G.clone()
    step()
    assert torch.allclose(m.flat.G, 2 * g1, rtol=2e-2, atol=1e-6)      # accumulates like autograd
    for p in m.parameters():
        p.grad = None
    step()
    assert torch.allclose(m.flat.G, g1, rtol=2e-2, atol=1e-6)          # set_to_none -> fresh gradients


def test_training_dropout_runs_and_is_deterministic(dev):
    m = _build("t5", torch.bfloat16, dev).train()
    mel, _, lab_pad, _ = _inputs(dev)
    lab = lab_pad[:, :256].contiguous()
    m.engine._stream_ctr = 0
    a = m(inputs=mel, labels=lab).detach().clone()
    m.engine._stream_ctr = 0
    b = m(inputs=mel, labels=lab).detach().clone()
    m.eval()
    with torch.no_grad():
        c = m(inputs=mel, labels=lab)
    assert torch.equal(a, b) and not torch.allclose(a, c, atol=1e-3)


@pytest.mark.parametrize("ml", [32, 256, 1024])
def test_greedy_t5_token_ids_bit_exact_fp32(dev, golden, ml):
    m = _build("t5", torch.float32, dev)
    mel = _inputs(dev)[0]
    ids = m.generate(mel, max_length=ml)
    assert m._decoder.graph_captured, "decode steps must replay from the captured hipGraph"
    np.testing.assert_array_equal(ids.cpu().numpy(), golden[f"t5.gen{ml}"])


@pytest.mark.parametrize("variant", ["segmem_v2", "segmem_v2_with_prev"])
@pytest.mark.parametrize("ml", [32, 256, 1024])
def test_greedy_segmem_token_ids_bit_exact_fp32(dev, golden, variant, ml):
    m = _build(variant, torch.float32, dev)
    mel = _inputs(dev)[0]
    ids = m.generate(mel, max_length=ml)
    assert ids.shape == (2, ml)
    np.testing.assert_array_equal(ids.cpu().numpy(), golden[f"{variant}.gen{ml}"])


@pytest.mark.parametrize("ml", [96, 256])
def test_greedy_segmem_v1_generate_2_bit_exact_fp32(dev, golden, ml):
    """T5SegMem.generate_2: memory rows prefill the self-attention cache, tokens start at position 64."""
    m = _build("segmem_v1", torch.float32, dev)
    mel = _inputs(dev)[0]
    ids = m.generate_2(mel, max_length=ml)
    assert ids.shape == (2, ml) and m._decoder.graph_captured
    np.testing.assert_array_equal(ids.cpu().numpy(), golden[f"segmem_v1.gen2_{ml}"])
    # and the plain batched decode of the same class
    np.testing.assert_array_equal(m.generate(mel, max_length=32).cpu().numpy(), golden["segmem_v1.gen32"])


def test_segmem_v1_generate_2_early_eos_matches_oracle(dev):
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    from oracle import t5_ref
    w = golden_weights(T5_SMALL, 1)
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][1] *= 3.2
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    mel = torch.from_numpy(synth_mel(3, seed=11))
    with torch.no_grad():
        ref = t5_ref.generate_segmem_v1(sd, T5_SMALL, mel, max_length=100)
    m = _build("segmem_v1", torch.float32, dev)
    with torch.no_grad():
        m.flat.load_numpy(w)
    ids = m.generate_2(mel.to(dev), max_length=100)
    assert (ref == 1).any(), "EOS never fired; raise the boost"
    assert torch.equal(ids.cpu(), ref)


def test_greedy_eos_handling_matches_oracle(dev):
    """Force early EOS: rows finish at different steps, later tokens are pad, loop stops early."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    from oracle import t5_ref
    w = golden_weights(T5_SMALL)
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][1] *= 3.2          # make EOS competitive so it wins within a few dozen steps
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    mel = torch.from_numpy(synth_mel(3, seed=11))
    with torch.no_grad():
        ref = t5_ref.generate_t5(sd, T5_SMALL, mel, max_length=200)
    m = _build("t5", torch.float32, dev)
    with torch.no_grad():
        m.flat.load_numpy(w)
    ids = m.generate(mel.to(dev), max_length=200, )
    assert ref.shape[1] < 201, "EOS never fired; raise the boost"
    assert torch.equal(ids.cpu(), ref)


def test_bf16_decode_runs_full_length(dev):
    m = _build("t5", torch.bfloat16, dev)
    with torch.no_grad():
        m.flat.master("lm_head.weight")[1].zero_()      # EOS never wins: all steps run
    mel = _inputs(dev)[0]
    ids = m.generate(mel, max_length=128)
    assert ids.shape == (2, 129) and (ids[:, 1:] != 1).all()


def test_lightning_style_task_step(dev):
    """tasks.mt3_net.MT3Net drives forward + torch CE + loss.backward() + torch AdamW like Lightning."""
    from mrmt3.synthetic import T5_SMALL
    from tasks.mt3_net import MT3Net
    optim_cfg = dict(lr=1e-3, warmup_steps=1, num_steps_per_epoch=10, num_epochs=1, min_lr=1e-4)
    task = MT3Net(dict(T5_SMALL, dropout_rate=0.0), optim_cfg)
    task.model.load_golden()
    task.to(dev).train()
    (opt,), (sched,) = task.configure_optimizers()
    mel, _, lab_pad, _ = _inputs(dev)
    lab = lab_pad[:, :128].contiguous()
    losses = []
    for it in range(6):
        opt.zero_grad()
        loss = task.training_step((mel, lab), it)
        loss.backward()
        opt.step()
        sched["scheduler"].step()
        losses.append(loss.item())
    assert "train_loss" in task.logged and all(np.isfinite(losses))
    assert losses[-1] < losses[0] - 0.05, losses     # lr 0 at step 0 (warm-up), then it learns


def test_greedy_large_batch_equals_small_groups(dev):
    """40 segments decoded as one group (5 argmax workgroups, batch on gridDim.y) == the same segments decoded
    8 at a time, early EOS included (the last-finishing workgroup folds the finished flags)."""
    import mrmt3.decode as dec_mod
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    w = golden_weights(T5_SMALL)
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][1] *= 3.2
    m = _build("t5", torch.float32, dev)
    with torch.no_grad():
        m.flat.load_numpy(w)
    mel = torch.from_numpy(synth_mel(40, seed=21)).to(dev)
    big = m.generate(mel, max_length=96)
    assert m._decoder.graph_captured
    old = dec_mod.MAX_DECODE_BATCH
    try:
        dec_mod.MAX_DECODE_BATCH = 8
        m._decoder = None
        small = m.generate(mel, max_length=96)
    finally:
        dec_mod.MAX_DECODE_BATCH = old
        m._decoder = None
    assert (big == 1).any(), "EOS never fired; raise the boost"
    assert torch.equal(big, small)


@pytest.mark.parametrize("dtype,n,group", [(torch.float32, 150, 64), (torch.bfloat16, 200, 100)])
def test_greedy_groups_beyond_64_sequences(dev, dtype, n, group):
    """The decoder takes up to 256 sequences per group (19 / 25 argmax workgroups, several MFMA column groups): the token
    ids of one large group equal those of the same segments decoded in smaller groups, early EOS included."""
    import mrmt3.decode as dec_mod
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    w = golden_weights(T5_SMALL)
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][1] *= 3.2
    m = _build("t5", dtype, dev)
    with torch.no_grad():
        m.flat.load_numpy(w)
    mel = torch.from_numpy(synth_mel(n, seed=33)).to(dev)
    if dtype == torch.bfloat16:
        mel = mel.bfloat16()
    big = m.generate(mel, max_length=48)
    assert big.shape[0] == n and dec_mod.MAX_DECODE_BATCH >= n
    old = dec_mod.MAX_DECODE_BATCH
    try:
        dec_mod.MAX_DECODE_BATCH = group
        m._decoder = None
        small = m.generate(mel, max_length=48)
    finally:
        dec_mod.MAX_DECODE_BATCH = old
        m._decoder = None
    assert torch.equal(big, small)


def test_bf16_large_batch_mfma_projections_agree_with_single_sequence_kernels(dev):
    """Batches > 8 run the decode projections on the matrix cores, 16 sequences per wave.  Both paths use
    bf16 operands and f32 accumulation, so they may differ only by the order of additions: over the first
    steps the token streams of 40 segments must coincide (a later near-tie may legitimately flip)."""
    import mrmt3.decode as dec_mod
    m = _build("t5", torch.bfloat16, dev)
    from mrmt3.synthetic import synth_mel
    mel = torch.from_numpy(synth_mel(40, seed=5)).to(dev)
    big = m.generate(mel, max_length=64)
    old = dec_mod.MAX_DECODE_BATCH
    try:
        dec_mod.MAX_DECODE_BATCH = 8
        m._decoder = None
        small = m.generate(mel, max_length=64)
    finally:
        dec_mod.MAX_DECODE_BATCH = old
        m._decoder = None
    n = min(big.shape[1], small.shape[1])
    assert torch.equal(big[:, :9], small[:, :9])
    # a flipped near-tie changes everything after it, so count how far each sequence stays identical
    eq = (big[:, :n] == small[:, :n]).long().cumprod(dim=1).sum(dim=1)
    print("identical prefix lengths:", sorted(eq.tolist()))
    assert (eq == n).float().mean().item() >= 0.6 and eq.min().item() >= 9, sorted(eq.tolist())


def test_long_context_finetune_shapes(dev):
    """BASELINE configs[4] (config_slakh_segmem_finetune.yaml): segmem_v2_with_prev with mel_length 2048, i.e.
    2048 encoder frames + 64 memory slots as cross-attention keys.  fp32 logits vs the oracle within 2e-4, bf16
    loss within 1e-3, bf16 gradients cosine > 0.995 for a sample of tensors."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
    from oracle import t5_ref
    torch.set_num_threads(8)
    B, Le, Ld = 1, 2048, 128
    mel = torch.from_numpy(synth_mel(B * 8, seed=31).reshape(B, Le, 512))
    lab = torch.from_numpy(synth_labels(B, Ld, full=False, seed=32, mean_len=90))
    prev = torch.from_numpy(synth_labels(B, Ld, full=False, seed=33, mean_len=90))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 1).items()}
    logits = t5_ref.forward_logits(sd, T5_SMALL, mel, lab, variant="segmem_v2_with_prev", targets_prev=prev.clone())
    ref_loss = t5_ref.ce_loss(logits, lab)
    ref_loss.backward()
    m32 = _build("segmem_v2_with_prev", torch.float32, dev)
    with torch.no_grad():
        got = m32(inputs=mel.to(dev), labels=lab.to(dev), targets_prev=prev.clone().to(dev))
    assert got.shape == (B, Ld, 1536)
    np.testing.assert_allclose(got.cpu().numpy(), logits.detach().numpy(), atol=2e-4, rtol=0)
    m = _build("segmem_v2_with_prev", torch.bfloat16, dev)        # eval mode: dropout off, autograd bridge on
    out = m(inputs=mel.to(dev), labels=lab.to(dev), targets_prev=prev.clone().to(dev))
    loss = torch.nn.functional.cross_entropy(out.view(-1, 1536), lab.to(dev).view(-1), ignore_index=-100)
    print("long context: bf16 loss %.5f oracle %.5f" % (loss.item(), ref_loss.item()))
    assert abs(loss.item() - ref_loss.item()) < 1e-3
    loss.backward()
    for key in ("encoder.block.0.layer.0.SelfAttention.q.weight", "encoder.block.7.layer.1.DenseReluDense.wo.weight",
                "decoder.block.0.layer.1.EncDecAttention.k.weight", "segmem_encoder.block.0.layer.0.SelfAttention.v.weight",
                "proj.weight", "lm_head.weight"):
        g, r = m.flat.grad(key).float().cpu().reshape(-1), sd[key].grad.reshape(-1)
        cos = torch.dot(g, r) / (g.norm() * r.norm())
        assert cos > 0.995, (key, cos.item())


@pytest.mark.parametrize("variant", ["segmem_v2_with_prev", "segmem_v2"])
def test_lockstep_decode_of_several_recordings_equals_one_at_a_time(dev, variant):
    """generate_songs: row s of the decode batch is recording s's current segment; every row must produce exactly
    what the sequential per-recording `generate` produces (fp32, early EOS and ragged segment counts included)."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    w = golden_weights(T5_SMALL, 1)
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][1] *= 3.0          # EOS fires at different steps in different rows
    m = _build(variant, torch.float32, dev)
    with torch.no_grad():
        m.flat.load_numpy(w)
    songs = [torch.from_numpy(synth_mel(n, seed=40 + n)).to(dev) for n in (3, 1, 2)]
    together = m.generate_songs(songs, max_length=80)
    assert [t.shape for t in together] == [(3, 80), (1, 80), (2, 80)]
    for s, mel in enumerate(songs):
        alone = m.generate(mel, max_length=80)
        assert torch.equal(together[s], alone), s
    assert any((t == 1).any().item() for t in together), "EOS never fired; raise the boost"


def test_segment_memory_length_zero_is_plain_mt3(dev):
    """test.sh's `model_segmem_length=0` experiment ("MR-MT3 with no memory block, which is basically MT3"): with an
    empty memory the V2WithPrev model's logits and greedy tokens equal the plain T5's on the same weights."""
    from mrmt3.synthetic import T5_SMALL, golden_weights
    from models.t5 import T5ForConditionalGeneration
    from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
    w = golden_weights(T5_SMALL, 1)
    m0 = T5SegMemV2WithPrev(T5_SMALL, segmem_num_layers=1, segmem_length=0, compute_dtype=torch.float32).to(dev).eval()
    t5 = T5ForConditionalGeneration(T5_SMALL, compute_dtype=torch.float32).to(dev).eval()
    with torch.no_grad():
        m0.flat.load_numpy(w)
        t5.flat.load_numpy({k: v for k, v in w.items() if not k.startswith("segmem")})
    mel, _, lab_pad, prev = _inputs(dev)
    with torch.no_grad():
        a = m0(inputs=mel, labels=lab_pad, targets_prev=prev.clone())
        b = t5(inputs=mel, labels=lab_pad)
    assert torch.allclose(a, b, atol=1e-5)
    ids0 = m0.generate(mel, max_length=48)
    for i in range(mel.shape[0]):                               # the segmem model decodes segment by segment, padded to max_length
        ref = t5.generate(mel[i:i + 1], max_length=48)[0]
        n = min(48, ref.numel())
        assert torch.equal(ids0[i, :n], ref[:n])
    # and the bf16 training step runs with an empty memory
    mb = T5SegMemV2WithPrev(T5_SMALL, segmem_num_layers=1, segmem_length=0).load_golden().to(dev)
    mb.train()
    out = mb(inputs=mel, labels=lab_pad[:, :128].contiguous(), targets_prev=prev[:, :128].clone())
    out.float().mean().backward()
    assert torch.isfinite(mb.flat.G).all()
