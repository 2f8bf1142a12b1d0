"""Round 6: the failed-capture abort of round 5 (gpurun_out/r5/s21_run2.log), hunted in ONE long-lived process.

Loop: a new Trainer (forced collectives at world 1 through the library's own RCCL communicator; in the round-5 tree, where this
first reproduced the abort at iteration ~55, with the collectives captured: MRMT3_DDP_GRAPH=1 / inline), 2 eager steps +
capture + 2 replays, then tear down in the order the round-5 tests used (buckets.close() before the graphs go).  Every k-th
trainer is kept alive so that graphs / streams / communicators pile up as they do in the GPU suite.  A capture that fails is
reported with its full state through MRMT3_CAPTURE_LOG; the process installs the native abort trace.
usage: r6_capture_stress.py [iterations] [seconds] [close_order: early|late|never]"""
import os
import socket
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mr-mt3_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 360.0
order = sys.argv[3] if len(sys.argv) > 3 else "early"
os.environ["MRMT3_DDP_FORCE_COLLECTIVES"] = "1"
os.environ["MRMT3_DDP_NATIVE"] = "1"
os.environ.setdefault("MRMT3_CAPTURE_LOG", os.path.join(ROOT, "gpurun_out", "r6", "capture_stress.log"))
os.makedirs(os.path.dirname(os.environ["MRMT3_CAPTURE_LOG"]), exist_ok=True)
from mrmt3 import lib
from mrmt3.synthetic import T5_SMALL, synth_audio, synth_labels
from mrmt3.trainer import Trainer
from models.t5 import T5ForConditionalGeneration

lib.abort_trace_install("")
dev = torch.device("cuda", 0)
s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
a = torch.from_numpy(synth_audio(8, seed=1)).to(dev)
t = torch.from_numpy(synth_labels(8, 256, full=False, seed=2, mean_len=90)).to(dev)
kept, fell_back, t0 = [], 0, time.time()
i = 0
for i in range(n_iter):
    if time.time() - t0 > budget:
        break
    mode = ""                                  # (round-5 tree: alternated MRMT3_DDP_GRAPH=1 / inline here; those forms are gone)
    m = T5ForConditionalGeneration(dict(T5_SMALL)).load_golden().to(dev)
    tr = Trainer(m, lr=1e-3, graph=True, layers_per_bucket=2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(5):
            tr.train_step(a, t, audio=True)
        torch.cuda.synchronize()
    msgs = [str(x.message) for x in w if "capture" in str(x.message) or "stream" in str(x.message)]
    if msgs or not tr.graph_captured:
        fell_back += 1
        print(f"iteration {i}: captured={tr.graph_captured} :: {msgs}", flush=True)
    if order == "early":
        tr.buckets.close()                     # the round-5 order: communicator destroyed, its captured nodes still alive
    if i % 7 == 0:
        kept.append(tr)                        # pile up graphs / streams / (with 'never') communicators
    elif order == "late":
        tr._graphs.clear()
        torch.cuda.synchronize()
        tr.buckets.close()
    if i % 25 == 0:
        print(f"iteration {i}: {time.time() - t0:.0f} s, {fell_back} fell back, kept {len(kept)}", flush=True)
print(f"done: {i + 1} iterations in {time.time() - t0:.0f} s, {fell_back} fell back (close order: {order})", flush=True)
dist.destroy_process_group()
