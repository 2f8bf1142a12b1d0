"""Golden-vector generator: imports the REFERENCE model code from /root/reference through a small
compatibility shim and records its outputs on seeded inputs.

Runs only in the build container (needs /root/reference; nothing here travels as source to the GPU
box except this script and the .npz files it writes).  Usage:

    python tests/golden/make_golden.py [--long]      # --long also records 1024-token decodes

The shim adapts the installed transformers 5.15 to the 4.18 API the reference was
written against; it does not change any arithmetic:
  1. modeling_t5.checkpoint        -> torch.utils.checkpoint.checkpoint (import fails otherwise)
  2. T5PreTrainedModel.get_head_mask(head_mask, n) -> [None]*n           (removed in 5.x)
  3. get_extended_attention_mask(mask, shape, device) with 4.18 semantics: causal for decoders,
     additive (1-m)*-10000.0
  4. invert_attention_mask with 4.18's (1-m)*-1e9
  5. config.tie_word_embeddings forced back to False after T5Config.from_dict (5.15 overrides it)
Weights are set explicitly from the golden recipe (never the default init: under 5.15 it breaks
the additive -10000 causal mask); eager attention; eval mode.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels  # noqa: E402


def install_shim():
    import transformers.models.t5.modeling_t5 as mt5
    import torch.utils.checkpoint
    from transformers import T5PreTrainedModel

    mt5.checkpoint = torch.utils.checkpoint.checkpoint

    def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
        return [None] * num_hidden_layers

    def get_extended_attention_mask(self, attention_mask, input_shape, device=None, dtype=None):
        # transformers 4.18 semantics
        if attention_mask.dim() == 3:
            ext = attention_mask[:, None, :, :]
        elif attention_mask.dim() == 2:
            if self.config.is_decoder:
                bsz, seq = input_shape
                ids = torch.arange(seq, device=attention_mask.device)
                causal = (ids[None, None, :].repeat(bsz, seq, 1) <= ids[None, :, None]).to(attention_mask.dtype)
                if causal.shape[1] < attention_mask.shape[1]:
                    pre = attention_mask.shape[1] - causal.shape[1]
                    causal = torch.cat([torch.ones((bsz, seq, pre), dtype=causal.dtype), causal], axis=-1)
                ext = causal[:, None, :, :] * attention_mask[:, None, None, :]
            else:
                ext = attention_mask[:, None, None, :]
        else:
            raise ValueError("bad mask")
        ext = ext.to(dtype=torch.float32)
        return (1.0 - ext) * -10000.0

    def invert_attention_mask(self, encoder_attention_mask):
        if encoder_attention_mask.dim() == 3:
            ext = encoder_attention_mask[:, None, :, :]
        else:
            ext = encoder_attention_mask[:, None, None, :]
        ext = ext.to(dtype=torch.float32)
        return (1.0 - ext) * -1e9

    T5PreTrainedModel.get_head_mask = get_head_mask
    T5PreTrainedModel.get_extended_attention_mask = get_extended_attention_mask
    T5PreTrainedModel.invert_attention_mask = invert_attention_mask


def build_reference(variant: str, segmem_length: int = 64):
    # the drop-in tree has same-named packages (`models`, `tasks`, ...): take it off the path so that
    # `models.*` can only resolve to the reference (checked below)
    sys.path[:] = [p for p in sys.path if not p.rstrip("/").endswith("mr-mt3_amd")]
    sys.path.insert(0, "/root/reference")
    from transformers import T5Config
    cfg = dict(T5_SMALL)
    cfg["use_cache"] = False
    tc = T5Config.from_dict(cfg)
    tc._attn_implementation = "eager"
    # 5th patch: transformers 5.15's T5Config forces tie_word_embeddings=True (and would tie
    # lm_head to decoder_embed_tokens and rescale by d_model^-0.5); under the pinned 4.18 the
    # reference's config (`tie_word_embeddings: false`, config/model/MT3Net.yaml:22) is honoured.
    tc.tie_word_embeddings = False
    if variant == "t5":
        from models.t5 import T5ForConditionalGeneration as M
        m = M(tc)
        seg_layers = 0
    else:
        mod = {"segmem_v1": ("models.t5_segmem", "T5SegMem"),
               "segmem_v2": ("models.t5_segmem_v2", "T5SegMemV2"),
               "segmem_v2_with_prev": ("models.t5_segmem_v2_with_prev", "T5SegMemV2WithPrev")}[variant]
        import importlib
        M = getattr(importlib.import_module(mod[0]), mod[1])
        m = M(tc, segmem_num_layers=1, segmem_length=segmem_length)
        seg_layers = 1
    assert sys.modules[M.__module__].__file__.startswith("/root/reference/"), sys.modules[M.__module__].__file__
    w = golden_weights(T5_SMALL, seg_layers)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # only aliases and inv_freq buffers may be missing
    bad = [k for k in missing if not (k.endswith("embed_tokens.weight") or k.endswith("inv_freq"))]
    assert not bad and not unexpected, (bad, unexpected)
    assert m.lm_head.weight.data_ptr() != m.decoder_embed_tokens.weight.data_ptr()
    assert not m.config.tie_word_embeddings
    for k, v in sd.items():   # every recipe tensor really is what the model holds
        assert torch.equal(m.state_dict()[k], v), k
    m.eval()
    return m


def zlib_seed(name):
    import zlib
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def sample_idx(shape, n, seed):
    rs = np.random.RandomState(seed)
    return rs.randint(0, int(np.prod(shape)), size=n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--lr", action="store_true", help="only record the reference's LR-schedule values")
    ap.add_argument("--param-order", action="store_true",
                    help="only record the reference models' parameters() order (optimizer-state indexing)")
    ap.add_argument("--bf16-bound", action="store_true",
                    help="only record what the REFERENCE itself loses under torch.autocast(bfloat16) vs its fp32 run "
                         "(tests/golden/bf16_bound.npz): the yardstick for the bf16 compute path's logit tolerance")
    ap.add_argument("--bench-shape", action="store_true",
                    help="only record the reference's fp32 loss / sampled logits / per-tensor gradient norms + samples, "
                         "and its autocast deviations, at B=16 x 1024 tokens (tests/golden/bench_shape.npz): the shape "
                         "at which the kernels the benchmark times dispatch")
    ap.add_argument("--long-shape", action="store_true",
                    help="only record the same quantities for segmem_v2_with_prev at BASELINE configs[4]'s shape — B=2 x 2048 "
                         "mel frames (+64 memory slots) x 1024 tokens (tests/golden/long_shape.npz)")
    ap.add_argument("--v1-decode", action="store_true",
                    help="only add T5SegMem's generate / generate_2 outputs to the existing npz")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shim()
    out = {}
    B = 2
    mel = torch.from_numpy(synth_mel(B))
    lab_full = torch.from_numpy(synth_labels(B, full=True))
    lab_pad = torch.from_numpy(synth_labels(B, full=False, seed=777))
    prev = torch.from_numpy(synth_labels(B, full=False, seed=999))
    import torch.nn.functional as F

    if args.lr:
        import importlib.util
        import json
        spec = importlib.util.spec_from_file_location("reference_utils", "/root/reference/utils.py")
        ru = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ru)
        rec = []
        for warm, total, min_lr in ((64500, 1289 * 800, 1e-4), (10, 100, 1e-4), (5, 50, 2e-5)):
            opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
            sch = ru.get_cosine_schedule_with_warmup(opt, warm, total, min_lr=min_lr)
            steps = sorted(set([0, 1, 2, warm // 2, warm - 1, warm, warm + 1, (warm + total) // 2, total - 1, total, total + 7]))
            lam = sch.lr_lambdas[0]
            rec.append({"num_warmup_steps": warm, "num_training_steps": total, "min_lr": min_lr, "steps": steps,
                        "multiplier": [float(lam(s)) for s in steps]})
        # the two unused schedulers of the same file, for completeness of the utils surface
        opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        noam = ru.get_noam_scheduler(opt, 4000, 512)
        noam_lrs = []
        for _ in range(6):
            noam_lrs.append(float(noam.get_last_lr()[0]))
            opt.step()
            noam.step()
        opt2 = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        lin = ru.get_mt3_optimizer(opt2, 8)
        rec = {"cosine": rec, "noam_first_lrs": noam_lrs,
               "mt3_linear_multiplier": [float(lin.lr_lambdas[0](s)) for s in range(12)]}
        with open(os.path.join(HERE, "lr_golden.json"), "w") as f:
            json.dump(rec, f, indent=0)
        print("wrote lr_golden.json", [len(r["steps"]) for r in rec["cosine"]])
        return
    if args.param_order:
        import json
        rec = {}
        for variant in ("t5", "segmem_v1", "segmem_v2", "segmem_v2_with_prev"):
            m = build_reference(variant)
            rec[variant] = {"parameters": [n for n, _ in m.named_parameters()],
                            "state_dict": list(m.state_dict().keys())}
        with open(os.path.join(HERE, "param_order.json"), "w") as f:
            json.dump(rec, f, indent=0)
        print({k: (len(v["parameters"]), len(v["state_dict"])) for k, v in rec.items()})
        return
    if args.bf16_bound:
        # The reference trains in fp32 (`trainer.precision: 32`); its only bf16 form is torch autocast, which casts the
        # operands of every Linear / matmul to bf16 and accumulates in f32 — the same operand arithmetic as the MFMA
        # path.  Its deviation from its own fp32 logits on the golden inputs is what bf16-operand arithmetic can reach
        # on this model; the HIP path is held to it (tests/test_model_gpu.py::test_bf16_logits_and_loss).
        rec = {}
        for variant in ("t5", "segmem_v1", "segmem_v2", "segmem_v2_with_prev"):
            m = build_reference(variant)
            with torch.no_grad():
                for tag, lab in (("full", lab_full), ("pad", lab_pad)):
                    kw = {"targets_prev": prev.clone()} if variant == "segmem_v2_with_prev" else {}
                    ref = m(inputs=mel, labels=lab, **kw)
                    kw = {"targets_prev": prev.clone()} if variant == "segmem_v2_with_prev" else {}
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        low = m(inputs=mel, labels=lab, **kw).float()
                    d = (low - ref)
                    valid = (lab.view(-1) != -100)
                    l_ref = F.cross_entropy(ref.view(-1, ref.shape[-1]).double(), lab.view(-1), ignore_index=-100).item()
                    l_low = F.cross_entropy(low.view(-1, low.shape[-1]).double(), lab.view(-1), ignore_index=-100).item()
                    idx = sample_idx(ref.shape, 4096, 1234)
                    rec[f"{variant}.{tag}.autocast_max_abs"] = np.float32(d.abs().max().item())
                    rec[f"{variant}.{tag}.autocast_max_abs_scored_rows"] = np.float32(
                        d.view(-1, d.shape[-1])[valid].abs().max().item())
                    rec[f"{variant}.{tag}.autocast_rel_l2"] = np.float32((d.norm() / ref.norm()).item())
                    rec[f"{variant}.{tag}.autocast_dloss"] = np.float64(l_low - l_ref)
                    rec[f"{variant}.{tag}.autocast_logit_val"] = low.reshape(-1)[idx].numpy().astype(np.float32)
                    rec[f"{variant}.{tag}.autocast_argmax_agree"] = np.float32(
                        (low.argmax(-1) == ref.argmax(-1)).float().mean().item())
                    print(variant, tag, {k.split(".")[-1]: float(v) for k, v in rec.items()
                                         if k.startswith(f"{variant}.{tag}.") and np.ndim(v) == 0}, flush=True)
        # gradients: same yardstick for the backward pass (inputs of tests/test_model_gpu.py::
        # test_bf16_gradients_vs_oracle_autograd): per-tensor rel-L2 / cosine of the autocast gradient vs the fp32 one
        lab_g = torch.from_numpy(synth_labels(B, 256, full=False, seed=777, mean_len=120))
        prev_g = torch.from_numpy(synth_labels(B, 256, full=False, seed=999, mean_len=120))
        for variant in ("t5", "segmem_v1", "segmem_v2", "segmem_v2_with_prev"):
            m = build_reference(variant)
            grads = {}
            for mode in ("fp32", "autocast"):
                m.zero_grad(set_to_none=True)
                kw = {"targets_prev": prev_g.clone()} if variant == "segmem_v2_with_prev" else {}
                if mode == "autocast":
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        lg = m(inputs=mel, labels=lab_g, **kw).float()
                else:
                    lg = m(inputs=mel, labels=lab_g, **kw)
                F.cross_entropy(lg.view(-1, lg.shape[-1]), lab_g.view(-1), ignore_index=-100).backward()
                grads[mode] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
            names, rels, coss = [], [], []
            for n, g32 in grads["fp32"].items():
                if g32.norm() == 0:
                    continue
                ga = grads["autocast"][n].float()
                names.append(n)
                rels.append(((ga - g32).norm() / g32.norm()).item())
                coss.append(F.cosine_similarity(ga.flatten(), g32.flatten(), dim=0).item())
            rec[f"{variant}.grad_names"] = np.array(names)
            rec[f"{variant}.grad_rel_l2"] = np.array(rels, dtype=np.float32)
            rec[f"{variant}.grad_cos"] = np.array(coss, dtype=np.float32)
            w = int(np.argmax(rels))
            print(variant, "autocast gradient: worst rel-L2 %.3e (%s), worst cos %.5f, median rel-L2 %.3e" % (
                rels[w], names[w], min(coss), float(np.median(rels))), flush=True)
        np.savez_compressed(os.path.join(HERE, "bf16_bound.npz"), **rec)
        print("wrote bf16_bound.npz")
        return
    if args.bench_shape or args.long_shape:
        from mrmt3.synthetic import bench_shape_inputs, long_shape_inputs
        mel_b, lab_b, prev_b = (torch.from_numpy(a) for a in (long_shape_inputs() if args.long_shape else bench_shape_inputs()))
        rec = {}
        for variant in (("segmem_v2_with_prev",) if args.long_shape else ("t5", "segmem_v2_with_prev")):
            m = build_reference(variant)
            grads, logits = {}, {}
            for mode in ("fp32", "autocast"):
                t0 = time.time()
                m.zero_grad(set_to_none=True)
                kw = {"targets_prev": prev_b.clone()} if variant == "segmem_v2_with_prev" else {}
                if mode == "autocast":
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        lg = m(inputs=mel_b, labels=lab_b, **kw).float()
                else:
                    lg = m(inputs=mel_b, labels=lab_b, **kw)
                loss = F.cross_entropy(lg.view(-1, lg.shape[-1]), lab_b.view(-1), ignore_index=-100)
                loss.backward()
                grads[mode] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
                logits[mode] = lg.detach()
                rec[f"{variant}.{mode}.loss"] = np.float64(F.cross_entropy(
                    lg.detach().view(-1, lg.shape[-1]).double(), lab_b.view(-1), ignore_index=-100).item())
                print(variant, mode, "loss %.6f  %.1fs" % (rec[f"{variant}.{mode}.loss"], time.time() - t0), flush=True)
            ref, low = logits["fp32"], logits["autocast"]
            idx = sample_idx(ref.shape, 8192, 4321)
            rec[f"{variant}.logit_idx"] = idx.astype(np.int64)
            rec[f"{variant}.logit_val"] = ref.reshape(-1)[idx].numpy().astype(np.float32)
            rec[f"{variant}.autocast_logit_val"] = low.reshape(-1)[idx].numpy().astype(np.float32)
            rec[f"{variant}.autocast_max_abs"] = np.float32((low - ref).abs().max().item())
            rec[f"{variant}.autocast_rel_l2"] = np.float32(((low - ref).norm() / ref.norm()).item())
            names, norms, rels, coss, samp_i, samp_v = [], [], [], [], [], []
            for n, g32 in grads["fp32"].items():
                ga = grads["autocast"][n].float()
                names.append(n)
                norms.append(g32.double().norm().item())
                rels.append(((ga - g32).norm() / g32.norm()).item() if g32.norm() > 0 else 0.0)
                coss.append(F.cosine_similarity(ga.flatten(), g32.flatten(), dim=0).item() if g32.norm() > 0 else 1.0)
                ii = sample_idx(g32.shape, 256, zlib_seed(n))
                samp_i.append(ii.astype(np.int64))
                samp_v.append(g32.reshape(-1)[ii].numpy().astype(np.float32))
            rec[f"{variant}.grad_names"] = np.array(names)
            rec[f"{variant}.grad_norm"] = np.array(norms, dtype=np.float64)
            rec[f"{variant}.grad_rel_l2"] = np.array(rels, dtype=np.float32)
            rec[f"{variant}.grad_cos"] = np.array(coss, dtype=np.float32)
            rec[f"{variant}.grad_sample_idx"] = np.stack(samp_i)
            rec[f"{variant}.grad_sample_val"] = np.stack(samp_v)
            w = int(np.argmax(rels))
            print(variant, "autocast gradient at B=%dx%d (%d frames): worst rel-L2 %.3e (%s), worst cos %.5f, median rel-L2 %.3e; "
                  "logits max|d| %.3e rel-L2 %.3e" % (lab_b.shape[0], lab_b.shape[1], mel_b.shape[1], rels[w], names[w], min(coss), float(np.median(rels)),
                                                        rec[f"{variant}.autocast_max_abs"], rec[f"{variant}.autocast_rel_l2"]),
                  flush=True)
        name = "long_shape.npz" if args.long_shape else "bench_shape.npz"
        np.savez_compressed(os.path.join(HERE, name), **rec)
        print("wrote", name)
        return
    if args.v1_decode:
        import contextlib, io
        path = os.path.join(HERE, "model_golden.npz")
        out = dict(np.load(path))
        m = build_reference("segmem_v1")
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):   # the reference prints every step
            for ml in (32, 256):
                out[f"segmem_v1.gen{ml}"] = m.generate(inputs=mel, max_length=ml).numpy().astype(np.int16)
            for ml in (96, 256):     # generate_2 asserts max_length >= segmem_length (t5_segmem.py:213)
                out[f"segmem_v1.gen2_{ml}"] = m.generate_2(inputs=mel, max_length=ml).numpy().astype(np.int16)
        np.savez_compressed(path, **out)
        print("added", [k for k in out if k.startswith("segmem_v1.gen")])
        return

    for variant in ("t5", "segmem_v1", "segmem_v2", "segmem_v2_with_prev"):
        t0 = time.time()
        m = build_reference(variant)
        with torch.no_grad():
            for tag, lab in (("full", lab_full), ("pad", lab_pad)):
                kw = {}
                if variant == "segmem_v2_with_prev":
                    kw["targets_prev"] = prev.clone()     # mutated in place by the reference
                logits = m(inputs=mel, labels=lab, **kw)
                loss = F.cross_entropy(logits.view(-1, logits.shape[-1]).double(), lab.view(-1), ignore_index=-100)
                idx = sample_idx(logits.shape, 4096, 1234)
                out[f"{variant}.{tag}.loss"] = np.float64(loss.item())
                out[f"{variant}.{tag}.logit_idx"] = idx.astype(np.int64)
                out[f"{variant}.{tag}.logit_val"] = logits.reshape(-1)[idx].numpy().astype(np.float32)
                out[f"{variant}.{tag}.argmax"] = logits.argmax(-1).numpy().astype(np.int16)
                out[f"{variant}.{tag}.logit_absmax"] = np.float32(logits.abs().max().item())
            if variant == "t5":
                # per-sublayer activations of a 1-layer slice (encoder block 0) for kernel-level tests
                x = m.proj(mel)
                out["t5.slice.proj"] = x[0, :8].numpy()
                enc = m.encoder(inputs_embeds=x, return_dict=True)[0]
                out["t5.slice.enc_out"] = enc[:, ::37, ::5].numpy()
            # greedy decode (reference algorithm, no KV cache)
            for ml in ((32, 256, 1024) if args.long else (32, 256)):
                if variant in ("segmem_v1",):
                    continue
                ids = m.generate(inputs=mel, max_length=ml)
                out[f"{variant}.gen{ml}"] = ids.numpy().astype(np.int16)
                print(variant, "gen", ml, ids.shape, "t=%.1fs" % (time.time() - t0), flush=True)
        print(variant, "done in %.1fs" % (time.time() - t0), flush=True)

    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "model_golden.npz"), {k: getattr(v, "shape", None) for k, v in out.items()})


def codec_golden():
    """contrib/event_codec.py has no third-party imports: record its tables directly."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ref_event_codec", "/root/reference/contrib/event_codec.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ranges = [m.EventRange('pitch', 0, 127), m.EventRange('velocity', 0, 1), m.EventRange('tie', 0, 0),
              m.EventRange('program', 0, 127), m.EventRange('drum', 0, 127)]     # vocabularies.py:118-139, bins=1
    c = m.Codec(max_shift_steps=1000, steps_per_second=100, event_ranges=ranges)
    idx = [0, 1, 999, 1000, 1001, 1060, 1128, 1129, 1130, 1131, 1132, 1200, 1259, 1260, 1300, 1387]
    out = {"num_classes": c.num_classes, "max_shift_steps": c.max_shift_steps,
           "decode": {str(i): [c.decode_event_index(i).type, c.decode_event_index(i).value] for i in idx},
           "ranges": {t: list(c.event_type_range(t)) for t in ("shift", "pitch", "velocity", "tie", "program", "drum")},
           "encode": {f"{t}:{v}": c.encode_event(m.Event(t, v)) for t, v in
                      (("shift", 7), ("pitch", 60), ("velocity", 1), ("tie", 0), ("program", 33), ("drum", 36))}}
    json.dump(out, open(os.path.join(HERE, "codec_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    codec_golden()
    main()
