"""Hydra-compatible composer (mrmt3.hydra_lite) on a config tree shaped like the reference's
(`config/config_slakh_segmem.yaml` + `config/model/*.yaml`): defaults list, `${a.b}` and
`${hydra:runtime.choices.model}` interpolation, `key=value` / `+key=value` / group overrides, and
`_target_` instantiation of the drop-in task classes (no GPU needed to construct them)."""
import textwrap

import pytest

from mrmt3 import hydra_lite

TOP = """
num_epochs: 800
devices: 1
model_type: ${hydra:runtime.choices.model}
dataset_type: ${hydra:runtime.choices.dataset}
seed: 365
path:
event_length: 1024
mel_length: 256
num_rows_per_batch: 12
model_segmem_length: 4
optim:
  lr: 2e-4
  warmup_steps: 64500
  num_epochs: ${num_epochs}
  num_steps_per_epoch: 1289
  min_lr: 1e-4
grad_accum: 1
trainer:
  precision: 32
  max_epochs: ${num_epochs}
  accumulate_grad_batches: ${grad_accum}
  strategy: "ddp_find_unused_parameters_false"
  devices: ${devices}
eval:
  eval_after_num_epoch: 400
  batch_size: 8
defaults:
  - model: MT3Net
  - dataset: Slakh
"""

MODEL = """
_target_: tasks.%s
config:
  architectures:
    - T5ForConditionalGeneration
  d_ff: 1024
  d_kv: 64
  d_model: 512
  decoder_start_token_id: 0
  dropout_rate: 0.1
  pad_token_id: 0
  eos_token_id: 1
  unk_token_id: 2
  feed_forward_proj: gated-gelu
  layer_norm_epsilon: 1e-06
  num_heads: 6
  num_decoder_layers: 8
  num_layers: 8
  tie_word_embeddings: false
  vocab_size: 1536
  %s
  use_cache: False
"""


@pytest.fixture()
def cfgdir(tmp_path):
    (tmp_path / "model").mkdir()
    (tmp_path / "dataset").mkdir()
    (tmp_path / "config_slakh_segmem.yaml").write_text(TOP)
    (tmp_path / "model" / "MT3Net.yaml").write_text(MODEL % ("mt3_net.MT3Net", ""))
    seg = "segmem_num_layers: 1\n  segmem_length: ${model_segmem_length}"
    (tmp_path / "model" / "MT3NetSegMemV2WithPrev.yaml").write_text(
        MODEL % ("mt3_net_segmem_v2_with_prev.MT3NetSegMemV2WithPrev", seg))
    (tmp_path / "dataset" / "Slakh.yaml").write_text("train:\n  mel_length: ${mel_length}\n")
    return str(tmp_path)


def test_compose_interpolation_and_overrides(cfgdir):
    cfg = hydra_lite.compose(cfgdir, "config_slakh_segmem", ["devices=[0,1]", "optim.lr=1e-5", "+eval.load_weights_strict=False"])
    assert cfg.model_type == "MT3Net" and cfg.dataset_type == "Slakh"
    assert cfg.optim.num_epochs == 800 and cfg.trainer.max_epochs == 800 and cfg.trainer.devices == [0, 1]
    assert cfg.optim.lr == 1e-5 and cfg.eval.load_weights_strict is False
    assert cfg.dataset.train.mel_length == 256
    with pytest.raises(KeyError):
        hydra_lite.compose(cfgdir, "config_slakh_segmem", ["optim.nope=1"])


def test_group_override_and_instantiate(cfgdir):
    cfg = hydra_lite.compose(cfgdir, "config_slakh_segmem", ["model=MT3NetSegMemV2WithPrev", "model_segmem_length=64"])
    assert cfg.model_type == "MT3NetSegMemV2WithPrev" and cfg.model.config.segmem_length == 64
    task = hydra_lite.instantiate(cfg.model, optim_cfg=cfg.optim, eval_cfg=cfg.eval)
    assert type(task).__name__ == cfg.model._target_.split(".")[-1]           # train.py:36
    assert task.model.segmem_length == 64 and task.model.flat.numel == 48519680
    (opt,), (sched,) = task.configure_optimizers()
    assert opt.defaults["lr"] == 2e-4 and sched["interval"] == "step"
    base = hydra_lite.instantiate(hydra_lite.compose(cfgdir, "config_slakh_segmem").model, optim_cfg=cfg.optim)
    assert base.model.flat.numel == 45896704 and len(base.model.state_dict()) == 193   # incl. 2 aliases + 2 inv_freq buffers


def test_lr_schedule_matches_oracle():
    from oracle import t5_ref
    from utils import cosine_warmup_lambda
    lam = cosine_warmup_lambda(64500, 1289 * 800, min_lr=1e-4)
    for s in (0, 1, 64499, 64500, 500000, 1031199, 1031200):
        assert lam(s) == t5_ref.cosine_lambda(s, 64500, 1289 * 800, min_lr=1e-4)


def test_lr_schedule_matches_reference_values():
    """utils.get_cosine_schedule_with_warmup against multipliers recorded from the reference's own utils.py
    (tests/golden/lr_golden.json, make_golden.py --lr): warm-up, the step at the boundary, the min_lr floor on the
    multiplier, and steps past the end (the cosine turns back up there, as in the reference)."""
    import json
    import os
    import torch
    from oracle import t5_ref
    from utils import cosine_warmup_lambda, get_cosine_schedule_with_warmup
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_golden.json")) as f:
        rec = json.load(f)
    noam_want, lin_want, rec = rec["noam_first_lrs"], rec["mt3_linear_multiplier"], rec["cosine"]
    from utils import get_mt3_optimizer, get_noam_scheduler, remove_state_dict_prefix
    o = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    noam, got = get_noam_scheduler(o, 4000, 512), []
    for _ in range(6):
        got.append(float(noam.get_last_lr()[0]))
        o.step()
        noam.step()
    assert got == noam_want
    lin = get_mt3_optimizer(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0), 8)
    assert [float(lin.lr_lambdas[0](s)) for s in range(12)] == lin_want
    assert list(remove_state_dict_prefix({"module.a.module.b": 1, "c": 2})) == ["a.b", "c"]
    for r in rec:
        lam = cosine_warmup_lambda(r["num_warmup_steps"], r["num_training_steps"], min_lr=r["min_lr"])
        for s, want in zip(r["steps"], r["multiplier"]):
            assert lam(s) == want, (r, s)
            assert t5_ref.cosine_lambda(s, r["num_warmup_steps"], r["num_training_steps"], min_lr=r["min_lr"]) == want
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    sch = get_cosine_schedule_with_warmup(opt, 10, 100, min_lr=1e-4)
    assert sch.get_last_lr()[0] == 0.0                       # step 0 of the warm-up
    opt.step()
    sch.step()
    assert abs(sch.get_last_lr()[0] - 2e-4 * rec[1]["multiplier"][1]) < 1e-18


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/config"), reason="reference checkout not present")
def test_the_reference_yaml_tree_composes_unchanged():
    """The reference's own config directory (read in place, never copied): every top-level config composes, every
    model group resolves `${...}` interpolations and instantiates the drop-in task class named by `_target_`, and the
    sanity check of train.py:36 (model_type == class name) holds."""
    import glob
    import os
    root = "/root/reference/config"
    tops = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(root, "*.yaml")))
    models = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(root, "model", "*.yaml")))
    assert "config_slakh_segmem" in tops and "MT3NetSegMemV2WithPrev" in models
    seen = set()
    for top in tops:
        cfg = hydra_lite.compose(root, top, [])
        assert cfg.trainer.precision == 32 and int(cfg.event_length) in (256, 1024) and float(cfg.optim.lr) > 0
        assert cfg.model_type == cfg.model._target_.split(".")[-1]
        assert isinstance(cfg.dataset.train.root_dir, str)
        # every key train.py / test.py read from the composed config
        assert int(cfg.dataloader.train.batch_size) >= 1 and int(cfg.num_rows_per_batch) >= 1 and int(cfg.mel_length) >= 256
        assert int(cfg.optim.warmup_steps) > 0 and int(cfg.optim.num_steps_per_epoch) > 0 and int(cfg.optim.num_epochs) > 0
        assert float(cfg.optim.min_lr) > 0 and "path" in cfg and "seed" in cfg
        ev = cfg.eval
        assert "audio_dir" in ev and "eval_dataset" in ev and "exp_tag_name" in ev       # batch_size etc. are optional (test.py defaults)
        assert isinstance(cfg.dataset.test.root_dir, str)
    for model in models:
        cfg = hydra_lite.compose(root, "config_slakh_segmem", [f"model={model}"])
        assert cfg.model_type == model
        task = hydra_lite.instantiate(cfg.model, optim_cfg=cfg.optim, eval_cfg=cfg.eval)
        assert type(task).__name__ == model
        assert task.model.cfg["vocab_size"] == 1536 and task.model.cfg["d_model"] == 512
        seen.add(type(task.model).__name__)
        conf = task.configure_optimizers()
        opt = conf[0][0] if isinstance(conf, tuple) else conf
        base_lr = opt.param_groups[0].get("initial_lr", opt.param_groups[0]["lr"])     # LambdaLR starts the warm-up at 0
        assert abs(base_lr - float(cfg.optim.lr)) < 1e-12
    assert {"T5ForConditionalGeneration", "T5SegMem", "T5SegMemV2", "T5SegMemV2WithPrev"} <= seen


@pytest.mark.skipif(not __import__("os").path.isfile("/root/reference/pretrained/config.json"), reason="reference checkout not present")
def test_builtin_t5_small_equals_the_reference_pretrained_config():
    """`InferenceHandler(model=None, weight_path=...)` builds the model from mrmt3.synthetic.T5_SMALL, the reference from
    pretrained/config.json (inference.py:36-44): the two must describe the same network."""
    import json
    from mrmt3.synthetic import T5_SMALL
    with open("/root/reference/pretrained/config.json") as f:
        ref = json.load(f)
    assert {k: T5_SMALL[k] for k in ref} == ref
    assert set(T5_SMALL) - set(ref) == {"use_cache"} and T5_SMALL["use_cache"] is False


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/config"), reason="reference checkout not present")
def test_the_reference_command_lines_compose():
    """The override lists of the reference's train.sh / test.sh: list literals, Hydra's own `hydra/...` groups (ignored),
    group selections, keys added with `+`, a value containing an escaped `=`, segmem_length 0."""
    root = "/root/reference/config"
    cfg = hydra_lite.compose(root, "config_slakh_segmem", [
        "devices=[0,1]", "hydra/job_logging=disabled", "model=MT3NetSegMemV2WithPrev", "dataset=SlakhPrev",
        "dataset_use_tf_spectral_ops=False", "dataset_is_randomize_tokens=True", "split_frame_length=2000",
        "model_segmem_length=64", "trainer.check_val_every_n_epoch=20", "eval.eval_after_num_epoch=400",
        "eval.eval_first_n_examples=3", "eval.eval_per_epoch=10", "eval.contiguous_inference=True"])
    assert cfg.devices == [0, 1] and cfg.trainer.devices == [0, 1] and cfg.model.config.segmem_length == 64
    assert cfg.dataset.train._target_.endswith("SlakhDatasetWithPrevSegmem") and cfg.dataset.train.split_frame_length == 2000
    assert cfg.eval.contiguous_inference is True and cfg.dataset.train.is_randomize_tokens is True
    cfg = hydra_lite.compose(root, "config_slakh_segmem", [
        "model=MT3NetSegMemV2WithPrev", "path=../../../pretrained/exp_segmemV2_prev_context\\=0.ckpt", "model_segmem_length=0",
        'eval.eval_dataset="Slakh"', "eval.exp_tag_name=slakh_mt3_official",
        "eval.audio_dir=/data/slakh2100_flac_redux/test/*/mix_16k.wav", "eval.midi_dir=/data/slakh2100_flac_redux/test/",
        "hydra/job_logging=disabled", "eval.is_sanity_check=True", "+eval.load_weights_strict=False"])
    assert cfg.path.endswith("exp_segmemV2_prev_context=0.ckpt") and cfg.model.config.segmem_length == 0
    assert cfg.eval.eval_dataset == "Slakh" and cfg.eval.load_weights_strict is False
    with pytest.raises(KeyError):
        hydra_lite.compose(root, "config_slakh_segmem", ["eval.no_such_key=1"])          # needs `+`, as in Hydra
