"""Where does the bf16 path's logit deviation come from?  
Runs the golden batch through the bf16 engine with (a) everything as shipped, (b) the final decoder norm emitted in f32
and lm_head computed by the exact-f32 MFMA GEMM on the f32 master weights, and compares both with the reference's
recorded fp32 logits and with the reference's own bf16-autocast deviation (tests/golden/bf16_bound.npz)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "mr-mt3_amd"), ROOT]
from mrmt3.synthetic import T5_SMALL, synth_mel, synth_labels
from models.t5 import T5ForConditionalGeneration

g = np.load(os.path.join(ROOT, "tests/golden/model_golden.npz"))
bound = np.load(os.path.join(ROOT, "tests/golden/bf16_bound.npz"))
dev = torch.device("cuda:0")
mel = torch.from_numpy(synth_mel(2)).to(dev)
lab = torch.from_numpy(synth_labels(2, full=False, seed=777)).to(dev)
idx = torch.from_numpy(g["t5.pad.logit_idx"]).to(dev)
ref = g["t5.pad.logit_val"]
for head in ("bf16", "f32"):
    m = T5ForConditionalGeneration(T5_SMALL).load_golden().to(dev).eval()
    m.engine.head_dtype = head
    with torch.no_grad():
        lg = m(inputs=mel, labels=lab)
    got = lg.reshape(-1)[idx].cpu().numpy()
    loss = torch.nn.functional.cross_entropy(lg.view(-1, 1536).double(), lab.view(-1), ignore_index=-100).item()
    print("lm_head %s: max|d| %.3e rel-L2 %.3e dloss %.2e   (reference autocast: max|d| %.3e rel-L2 %.3e)" % (
        head, np.abs(got - ref).max(), np.linalg.norm(got - ref) / np.linalg.norm(ref), loss - float(g["t5.pad.loss"]),
        float(bound["t5.pad.autocast_max_abs"]), float(bound["t5.pad.autocast_rel_l2"])))
