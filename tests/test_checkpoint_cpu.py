"""Checkpoint interop — host logic, CPU only.

  * `reference_parameter_order` against the order recorded from the reference models themselves
    (tests/golden/param_order.json, written by make_golden.py --param-order);
  * the T5X key map of tools/convert_weight.py: round trip, kernel transposes, alias fill;
  * the three file forms the reference reads (`.pt` bare, `.pth` with `model.` keys, Lightning `.ckpt`);
  * the AdamW state layout against a real torch.optim.AdamW (load_state_dict / state_dict round trip).
"""
import json
import os

import numpy as np
import pytest
import torch

from mrmt3 import checkpoint as ck
from mrmt3.params import FlatParams
from mrmt3.synthetic import T5_SMALL, golden_weights

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {"t5": 0, "segmem_v1": 1, "segmem_v2": 1, "segmem_v2_with_prev": 1}


@pytest.fixture(scope="module")
def recorded():
    with open(os.path.join(HERE, "golden", "param_order.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_parameter_order_matches_reference(recorded, variant):
    order = ck.reference_parameter_order(T5_SMALL, VARIANTS[variant])
    assert order == recorded[variant]["parameters"]
    # and the state-dict key set (aliases + inv_freq buffers included) is the reference's
    assert set(_module(variant).state_dict()) == set(recorded[variant]["state_dict"])


def test_t5x_key_map_round_trip():
    from tools.convert_weight import convert_t5x_to_pt, pt_to_t5x, t5x_key_map
    w = golden_weights(T5_SMALL)
    sd = dict(w)
    sd["encoder.embed_tokens.weight"] = w["proj.weight"]
    sd["decoder.embed_tokens.weight"] = w["decoder_embed_tokens.weight"]
    flat = pt_to_t5x(T5_SMALL, sd)
    # Flax Dense kernels are [in, out]
    assert flat["target/encoder/layers_0/attention/query/kernel"].shape == (512, 384)
    assert flat["target/decoder/layers_7/mlp/wo/kernel"].shape == (1024, 512)
    assert flat["target/decoder/logits_dense/kernel"].shape == (512, 1536)
    assert flat["target/decoder/token_embedder/embedding"].shape == (1536, 512)
    assert len(flat) == len(set(v[0] for v in t5x_key_map(T5_SMALL).values())) == 189
    flat["state/step"] = np.zeros(())          # passed through untouched, like the reference's mapper
    out = convert_t5x_to_pt(T5_SMALL, flat)
    assert "state/step" in out and not any(k.startswith("target/") for k in out)
    for k, v in sd.items():
        assert torch.equal(out[k], torch.from_numpy(np.asarray(v))), k
    assert torch.equal(out["encoder.embed_tokens.weight"], out["proj.weight"])


def _module(variant="t5"):
    import importlib
    mod, cls = {"t5": ("models.t5", "T5ForConditionalGeneration"),
                "segmem_v1": ("models.t5_segmem", "T5SegMem"),
                "segmem_v2": ("models.t5_segmem_v2", "T5SegMemV2"),
                "segmem_v2_with_prev": ("models.t5_segmem_v2_with_prev", "T5SegMemV2WithPrev")}[variant]
    M = getattr(importlib.import_module(mod), cls)
    return M(dict(T5_SMALL)) if variant == "t5" else M(dict(T5_SMALL), segmem_num_layers=1, segmem_length=64)


def test_three_file_forms(tmp_path):
    m = _module().load_golden()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    torch.save(sd, tmp_path / "last.pt")
    torch.save({"model." + k: v for k, v in sd.items()}, tmp_path / "w.pth")
    torch.save(ck.lightning_checkpoint(m), tmp_path / "last.ckpt")
    for name in ("last.pt", "w.pth", "last.ckpt"):
        got = ck.read_checkpoint(str(tmp_path / name))
        assert list(got["state_dict"]) == list(sd), name
        fresh = _module()
        missing, unexpected = fresh.load_state_dict(got["state_dict"], strict=True)
        assert not missing and not unexpected
        assert torch.equal(fresh.flat.P, m.flat.P), name
    blob = torch.load(tmp_path / "last.ckpt", weights_only=False)
    assert all(k.startswith("model.") for k in blob["state_dict"])
    with pytest.raises(ValueError):
        ck.read_checkpoint(str(tmp_path / "weights.bin"))


def test_load_from_checkpoint_classmethod(tmp_path):
    from tasks.mt3_net import MT3Net
    m = _module().load_golden()
    torch.save(ck.lightning_checkpoint(m), tmp_path / "a.ckpt")
    optim_cfg = dict(lr=2e-4, warmup_steps=10, num_steps_per_epoch=10, num_epochs=1, min_lr=1e-4)
    task = MT3Net.load_from_checkpoint(str(tmp_path / "a.ckpt"), config=dict(T5_SMALL), optim_cfg=optim_cfg)   # test.py:98-102
    assert torch.equal(task.model.flat.P, m.flat.P)


@pytest.mark.parametrize("variant", ["t5", "segmem_v2_with_prev"])
def test_adamw_state_layout_against_torch_optimizer(variant):
    seg = VARIANTS[variant]
    m = _module(variant)
    flat = m.flat
    flat.ensure_adam()
    g = torch.Generator().manual_seed(3)
    flat.M.copy_(torch.randn(flat.numel, generator=g))
    flat.V.copy_(torch.rand(flat.numel, generator=g))
    order = ck.reference_parameter_order(T5_SMALL, seg)
    assert sorted(order) == sorted(flat.shapes)           # every owned tensor exactly once
    state = ck.adamw_state_from_flat(flat, order, step=7, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    # a real AdamW over the parameters in the reference's order accepts it ...
    params = [m._views[k] for k in order]
    mine = list(m.parameters())          # the drop-in module enumerates its parameters in the same order,
    assert len(mine) == len(params) and all(a is b for a, b in zip(mine, params))   # so Lightning's optimizer agrees
    opt = torch.optim.AdamW(params, lr=2e-4)
    opt.load_state_dict(state)
    assert opt.param_groups[0]["lr"] == 1e-4
    assert torch.equal(opt.state[params[5]]["exp_avg"], flat.view(flat.M, order[5]))
    # ... and what it writes back lands in a second flat store unchanged
    f2 = FlatParams(dict(T5_SMALL, num_decoder_layers=8), seg)
    step = ck.adamw_state_to_flat(opt.state_dict(), f2, order)
    assert step == 7 and torch.equal(f2.M, flat.M) and torch.equal(f2.V, flat.V)
    bad = {"state": {}, "param_groups": [{"params": [0, 1]}]}
    with pytest.raises(ValueError):
        ck.adamw_state_to_flat(bad, f2, order)


def test_lightning_checkpoint_lr_matches_what_torch_lambdalr_holds_after_n_steps():
    """the saved lr / _last_lr must be lambda(N) (what torch's LambdaLR holds after N optimizer steps and
    the resumed step uses), not the lr of the last executed step; and there is no empty `loops` dict that
    Lightning's restore_loops() would index with 'fit_loop'."""
    from types import SimpleNamespace
    from utils import cosine_warmup_lambda
    m = _module().load_golden()
    m.flat.ensure_adam()
    lam = cosine_warmup_lambda(10, 100, min_lr=1e-4)
    n = 7
    params = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.AdamW(params, lr=2e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    for _ in range(n):
        opt.step()
        sched.step()
    fake = SimpleNamespace(lr_dev=torch.tensor([2e-4 * lam(n - 1)]), host_step=n, base_lr=2e-4, lr_lambda=lam,
                           betas=(0.9, 0.999), eps=1e-8, wd=0.01)
    blob = ck.lightning_checkpoint(m, fake)
    assert "loops" not in blob
    assert blob["optimizer_states"][0]["param_groups"][0]["lr"] == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12)
    assert blob["lr_schedulers"][0]["_last_lr"][0] == pytest.approx(sched.get_last_lr()[0], rel=1e-12)
    assert blob["lr_schedulers"][0]["last_epoch"] == sched.state_dict()["last_epoch"]
    assert blob["lr_schedulers"][0]["_step_count"] == sched.state_dict()["_step_count"]
