"""Greedy decode of N tokens at batch B (bf16, EOS disabled) for per-kernel profiling:
   rocprofv3 --kernel-trace --stats -- python3 profiles/tools/decode_micro.py 8 256"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib
if os.environ.get("MRMT3_TOOL_LIB"):      # tuning tool only: A/B a variant build of the library
    lib.LIB_PATH = os.environ["MRMT3_TOOL_LIB"]
from mrmt3.synthetic import T5_SMALL, synth_mel
from models.t5 import T5ForConditionalGeneration

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
m = T5ForConditionalGeneration(T5_SMALL, compute_dtype=torch.bfloat16).load_golden().to(dev).eval()
with torch.no_grad():
    m.flat.master("lm_head.weight")[1].zero_()
mel = torch.from_numpy(synth_mel(B)).to(dev)
m.generate(mel, max_length=N)
torch.cuda.synchronize()
t0 = time.perf_counter()
ids = m.generate(mel, max_length=N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={B} tokens={N}: {dt*1e3:.1f} ms, {dt/N*1e6:.1f} us/step, graph={m._decoder.graph_captured}")
