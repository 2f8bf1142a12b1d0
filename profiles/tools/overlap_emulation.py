"""What does the bucketed gradient exchange cost the training step at N GPUs — measured on ONE GPU with emulated collectives.

The data-parallel step (mrmt3/trainer.py, the default form: one graph per gradient bucket, the bucket's all-reduce eager on the
collective stream between two replays, AdamW behind the last one) runs unchanged; only the all-reduce itself is replaced by
`mrmt3_comm_emulate` (diagnostics library): a kernel on the collective stream that occupies `CTAS` CUs, reads and rewrites the
bucket in place, and lasts as long as a ring all-reduce of the bucket would at an assumed BUS bandwidth:
    t = 2 (N - 1) / N * bytes / busbw.
So the step pays what it would pay at N GPUs for everything that is local to a GPU: the hardware-queue sharing, the second
busy queue's cost per dependent launch (profiles/r05_collectives_ab.txt), the CUs and memory bandwidth the collective
takes, the last bucket's exposed tail — and nothing for what is not: launch skew between ranks, stragglers, link-level effects.
efficiency = t(plain step) / t(step with the emulated exchange) is an UPPER bound of the weak-scaling efficiency under that
bandwidth assumption.
    python3 profiles/tools/overlap_emulation.py [steps = 20]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
os.environ["MRMT3_DDP_FORCE_COLLECTIVES"] = "1"
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from mrmt3 import lib, ddp
from mrmt3.synthetic import synth_audio, synth_labels
from mrmt3.trainer import Trainer
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
L = lib.load()
L.mrmt3_comm_emulate.restype = ctypes.c_int
L.mrmt3_comm_emulate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29555", rank=0, world_size=1)
CFG = {"n": 1, "busbw": 300e9, "ctas": 32}


def emulated_all_reduce(self, t, stream=None):
    """GradBuckets._all_reduce: the same stream discipline, the collective replaced by its stand-in."""
    s = stream
    if s is None:
        s = self._launch_stream(t.device)
        s.wait_stream(torch.cuda.current_stream(t.device))
    n = CFG["n"]
    seconds = 0.0 if n <= 1 else 2.0 * (n - 1) / n * t.numel() * t.element_size() / CFG["busbw"]
    rc = L.mrmt3_comm_emulate(ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size() // 4, seconds, CFG["ctas"], ctypes.c_void_p(s.cuda_stream))   # (words of 4 bytes)
    assert rc == 0, L.mrmt3_last_error()
    ev = torch.cuda.Event()
    ev.record(s)
    return ddp._StreamWork(ev, t.device)


real_all_reduce = ddp.GradBuckets._all_reduce


def step_ms(variant, B, n, busbw, plain=False, form=""):
    CFG.update(n=n, busbw=busbw)
    os.environ["MRMT3_DDP_GRAPH"] = form
    if plain:
        os.environ["MRMT3_DDP_FORCE_COLLECTIVES"] = "0"
    else:
        os.environ["MRMT3_DDP_FORCE_COLLECTIVES"] = "1"
        ddp.GradBuckets._all_reduce = emulated_all_reduce
    try:
        m = bench.build_model(variant, dev)
        tr = Trainer(m, lr=2e-4)
        audio = torch.from_numpy(synth_audio(B, seed=365)).to(dev)
        lab = torch.from_numpy(synth_labels(B, seed=365)).to(dev)
        prev = torch.from_numpy(synth_labels(B, seed=1365)).to(dev) if variant != "t5" else None
        f = lambda: tr.train_step(audio, lab, None if prev is None else prev.clone(), audio=True)
        while tr.use_graph and not tr.graph_captured:
            f()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        nb, mb = len(tr.buckets.buckets), [round((b["end"] - b["start"]) * 4 / 1e6) for b in tr.buckets.buckets]
        del tr, m
        torch.cuda.empty_cache()
        return dt, nb, mb
    finally:
        ddp.GradBuckets._all_reduce = real_all_reduce


print("emulated gradient exchange on one GPU: step ms (graph replays, %d steps), efficiency = plain / emulated; collective stand-in on %d CUs; library %d"
      % (steps, CFG["ctas"], L.mrmt3_version()))
CONFIGS = [(c.split(":")[0], int(c.split(":")[1])) for c in
           os.environ.get("EMU_CONFIGS", "t5:64,t5:12,segmem_v2_with_prev:64,segmem_v2_with_prev:12").split(",")]
for variant, B in CONFIGS:
    plain, _, _ = step_ms(variant, B, 1, 300e9, plain=True)
    zero, nb, mb = step_ms(variant, B, 1, 300e9)                    # the exchange's structure alone: zero-length collectives
    print("%s, %d segments per GPU: plain step %.3f ms; %d buckets of %s MB; segments with zero-length collectives %.3f ms (%.3f)"
          % (variant, B, plain, nb, mb, zero, plain / zero))
    for busbw in (300e9, 150e9):
        row = []
        for n in (2, 4, 8):
            t, _, _ = step_ms(variant, B, n, busbw)
            row.append("N=%d: %.3f ms (%.3f)" % (n, t, plain / t))
        print("    bus bandwidth %3.0f GB/s   %s" % (busbw / 1e9, "   ".join(row)))
dist.destroy_process_group()
