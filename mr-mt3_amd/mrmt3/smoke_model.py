"""Tiny end-to-end invocation of the model path used by `__graft_entry__.smoke()`: one forward +
backward of MT3Net-sized T5 on a 1-segment batch and a short greedy decode, checked against the
oracle (test infrastructure) on the same seeded inputs."""
import numpy as np
import torch


def run():
    from models.t5 import T5ForConditionalGeneration
    from mrmt3 import lib
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
    from oracle import t5_ref
    dev = torch.device("cuda:0")
    mel = torch.from_numpy(synth_mel(1))
    lab = torch.from_numpy(synth_labels(1, full=False, seed=777))
    sd = {k: torch.from_numpy(v) for k, v in golden_weights(T5_SMALL).items()}
    with torch.no_grad():
        ref = t5_ref.forward_logits(sd, T5_SMALL, mel, lab)
        ref_loss = t5_ref.ce_loss(ref, lab).item()
    model = T5ForConditionalGeneration(T5_SMALL).load_golden().to(dev).eval()
    logits = model(inputs=mel.to(dev), labels=lab.to(dev))
    loss = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), lab.to(dev).view(-1), ignore_index=-100)
    loss.backward()
    err = (logits.detach().cpu() - ref).abs().max().item()
    gnorm = model.flat.G.norm().item()
    assert abs(loss.item() - ref_loss) < 2e-3 and err < 5e-2 and np.isfinite(gnorm) and gnorm > 0, (loss.item(), ref_loss, err, gnorm)
    print("smoke: bf16 fwd+bwd loss %.5f (oracle %.5f) max|dlogit| %.3e |grad| %.3e" % (loss.item(), ref_loss, err, gnorm))
    m32 = T5ForConditionalGeneration(T5_SMALL, compute_dtype=torch.float32).load_golden().to(dev).eval()
    ids = m32.generate(mel.to(dev), max_length=16).cpu()
    with torch.no_grad():
        ref_ids = t5_ref.generate_t5(sd, T5_SMALL, mel, max_length=16)
    assert torch.equal(ids, ref_ids), (ids, ref_ids)
    print("smoke: fp32 greedy decode of 16 tokens matches the oracle:", ids[0, :8].tolist(), "...")
