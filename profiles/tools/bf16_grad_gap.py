"""Per-tensor gradient deviation of the bf16 HIP path vs the fp32 oracle, next to what the REFERENCE loses under
bf16 autocast on the same inputs (tests/golden/bf16_bound.npz).  Usage: python profiles/tools/bf16_grad_gap.py [variant]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "mr-mt3_amd"), ROOT]
from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
from oracle import t5_ref

variant = sys.argv[1] if len(sys.argv) > 1 else "t5"       # t5 | segmem_v2_with_prev
bound = np.load(os.path.join(ROOT, "tests/golden/bf16_bound.npz"))
ref_rel = dict(zip(bound[f"{variant}.grad_names"].tolist(), bound[f"{variant}.grad_rel_l2"].tolist()))
torch.set_num_threads(16)
B = 2
mel = torch.from_numpy(synth_mel(B))
lab = torch.from_numpy(synth_labels(B, 256, full=False, seed=777, mean_len=120))
prev = torch.from_numpy(synth_labels(B, 256, full=False, seed=999, mean_len=120))
sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 0 if variant == "t5" else 1).items()}
t5_ref.ce_loss(t5_ref.forward_logits(sd, T5_SMALL, mel, lab, variant=variant, targets_prev=prev.clone()), lab).backward()
dev = torch.device("cuda:0")
if variant == "t5":
    from models.t5 import T5ForConditionalGeneration as M
    m = M(T5_SMALL)
else:
    from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev as M
    m = M(T5_SMALL, segmem_num_layers=1, segmem_length=64)
m = m.load_golden().to(dev).eval()
out = m(inputs=mel.to(dev), labels=lab.to(dev), targets_prev=prev.clone().to(dev))
torch.nn.functional.cross_entropy(out.view(-1, 1536), lab.to(dev).view(-1), ignore_index=-100).backward()
rows = []
for k, ref in sd.items():
    if ref.grad is None or ref.grad.norm() == 0:
        continue
    g = m.flat.grad(k).cpu()
    rows.append((((g - ref.grad).norm() / ref.grad.norm()).item(), ref_rel.get(k, float("nan")), k))
rows.sort(reverse=True)
print("HIP rel-L2   reference-autocast rel-L2   tensor")
for r in rows[:25]:
    print("%.3e   %.3e   %s" % r)
print("median HIP %.3e, median autocast %.3e" % (np.median([r[0] for r in rows]), np.nanmedian([r[1] for r in rows])))
import collections
fam = collections.defaultdict(list)
for a, b, k in rows:
    fam[k.split(".")[-2] + ("." + k.split(".")[-3] if "Attention" in k else "")].append((a, b))
for f, v in sorted(fam.items()):
    print("%-32s HIP mean %.3e  autocast mean %.3e  (n=%d)" % (f, np.mean([x[0] for x in v]), np.nanmean([x[1] for x in v]), len(v)))
