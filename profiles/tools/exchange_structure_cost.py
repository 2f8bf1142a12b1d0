"""Where do the 0.3-0.45 ms go that the bucketed exchange costs the step BEFORE any byte is exchanged?  The step with forced
collectives (world 1), with the all-reduce (a) as a zero-length stand-in kernel on the collective stream + the usual stream waits and
events, (b) skipped together with its stream waits and events (only the graph segments and the per-bucket weight-gradient launches are
left), against the plain step.    python3 profiles/tools/exchange_structure_cost.py [steps = 30]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from mrmt3 import lib, ddp
from mrmt3.synthetic import synth_audio, synth_labels
from mrmt3.trainer import Trainer
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
L = lib.load()
L.mrmt3_comm_emulate.restype = ctypes.c_int
L.mrmt3_comm_emulate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29556", rank=0, world_size=1)
real_fire, real_ar = ddp.GradBuckets._fire, ddp.GradBuckets._all_reduce


def zero_all_reduce(self, t, stream=None):
    s = stream
    if s is None:
        s = self._launch_stream(t.device)
        s.wait_stream(torch.cuda.current_stream(t.device))
    assert L.mrmt3_comm_emulate(ctypes.c_void_p(t.data_ptr()), 1024, 0.0, 1, ctypes.c_void_p(s.cuda_stream)) == 0
    ev = torch.cuda.Event()
    ev.record(s)
    return ddp._StreamWork(ev, t.device)


def no_fire(self, idx):
    if idx in self._fired:
        return
    self._fired.add(idx)
    if self.before_fire is not None:      # (the per-bucket weight-gradient launch stays: eager and captured steps must plan alike)
        self.before_fire()


def step_ms(B, mode):
    os.environ["MRMT3_DDP_FORCE_COLLECTIVES"] = "0" if mode == "plain" else "1"
    ddp.GradBuckets._all_reduce = zero_all_reduce if mode == "zero" else real_ar
    ddp.GradBuckets._fire = no_fire if mode == "segments_only" else real_fire
    try:
        m = bench.build_model("t5", dev)
        tr = Trainer(m, lr=2e-4)
        audio = torch.from_numpy(synth_audio(B, seed=365)).to(dev)
        lab = torch.from_numpy(synth_labels(B, seed=365)).to(dev)
        f = lambda: tr.train_step(audio, lab, audio=True)
        while tr.use_graph and not tr.graph_captured:
            f()
        assert tr.graph_captured, "the step did not capture"
        for _ in range(3):
            f()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                f()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        n = len(next(iter(tr._graphs.values())).segments) + 1
        del tr, m
        torch.cuda.empty_cache()
        return best, n
    finally:
        ddp.GradBuckets._all_reduce, ddp.GradBuckets._fire = real_ar, real_fire


print("structure cost of the bucketed exchange, MT3Net, world 1 (best of 3 x %d steps); library %d" % (steps, L.mrmt3_version()))
for B in (64, 12):
    for lpb in ("4", "8"):
        os.environ["MRMT3_DDP_LAYERS_PER_BUCKET"] = lpb
        p, _ = step_ms(B, "plain")
        so, n = step_ms(B, "segments_only")
        z, _ = step_ms(B, "zero")
        print("%2d segments, %s layers per bucket (%d graphs per step): plain %.3f ms | graph segments + per-bucket weight-gradient launches only %.3f (+%.3f)"
              " | + stream waits, events and a zero-length kernel on the collective stream %.3f (+%.3f)" % (B, lpb, n, p, so, so - p, z, z - so))
dist.destroy_process_group()
