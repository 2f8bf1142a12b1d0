"""Can the collective stream wait for "bucket j complete" WITHOUT a resident spinning kernel?  hipStreamWaitValue32 makes the
command processor wait until a word of signal memory reaches a value.  (1) Does it order a stream behind a kernel of another
stream that adds to that word (mrmt3_flag_signal)?  (2) Does a stream parked in such a wait slow the compute graph down the way a
spinning kernel does (+9 us per dependent launch, profiles/r05_collectives_ab.txt)?
    python3 profiles/tools/wait_value_probe.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MRMT3_TOOL_LIB", os.path.join(ROOT, "mr-mt3_amd", "mrmt3", "libmrmt3_hip_diag.so"))
MODE = sys.argv[1] if len(sys.argv) > 1 else "write"     # who signals: "write" hipStreamWriteValue32, "sys" / "agent" a kernel's atomic add
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
sys.path.insert(0, ROOT)
import torch
from mrmt3 import lib
from mrmt3.synthetic import synth_audio, synth_labels
from mrmt3.trainer import Trainer
import bench

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
L = lib.load()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
L.mrmt3_flag_signal_sys.restype, L.mrmt3_flag_signal_sys.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]
print("signal by:", MODE, flush=True)


def signal(stream, target):
    if MODE == "write":
        assert hip.hipStreamWriteValue32(ctypes.c_void_p(stream.cuda_stream), sig, target, 0) == 0
    elif MODE == "sys":
        assert L.mrmt3_flag_signal_sys(sig, ctypes.c_void_p(stream.cuda_stream)) == 0
    else:
        assert L.mrmt3_flag_signal(sig, ctypes.c_void_p(stream.cuda_stream)) == 0
hipMallocSignalMemory, GTE = 0x2, 0
for name, flags in (("hipMallocSignalMemory", hipMallocSignalMemory),):
    sig = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, flags)
    print("hipExtMallocWithFlags(%s): rc %d ptr %#x" % (name, rc, sig.value or 0))
    assert rc == 0
hip.hipMemset(sig, 0, 8)
torch.cuda.synchronize()


def value():
    out = ctypes.c_uint32(0)
    hip.hipMemcpy(ctypes.byref(out), sig, 4, 2)
    return out.value


a, b = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)      # b at HIGH priority: a queue pool of its own (a normal-priority b shared the default
                                                                   # stream's hardware queue in the first run: part (2) deadlocked behind its own wait)
x = torch.zeros(1 << 24, device=dev)
y = torch.zeros(4, device=dev)
ok = True
for rnd in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = hip.hipStreamWaitValue32(ctypes.c_void_p(b.cuda_stream), sig, rnd + 1, GTE, 0xFFFFFFFF)
    assert rc == 0, rc
    with torch.cuda.stream(b):
        y[rnd] = x[-1]
    with torch.cuda.stream(a):
        for _ in range(20):
            x.add_(0.0)
        x.fill_(float(rnd + 1))
        signal(a, rnd + 1)
    torch.cuda.synchronize()
    print("round %d: %.2f ms, signal word %d, consumer saw %.0f" % (rnd, (time.perf_counter() - t0) * 1e3, value(), y[rnd].item()), flush=True)
    ok = ok and y[rnd].item() == rnd + 1
print("(1) a stream parked in hipStreamWaitValue32 is released by a KERNEL's device-scope atomic add on the signal word:", "yes" if ok else "NO")

# (2) the compute graph beside a stream parked in a wait for the whole step
m = bench.build_model("t5", dev)
tr = Trainer(m, lr=2e-4)
audio = torch.from_numpy(synth_audio(64, seed=365)).to(dev)
lab = torch.from_numpy(synth_labels(64, seed=365)).to(dev)
while not tr.graph_captured:
    tr.train_step(audio, lab, audio=True)
cur = torch.cuda.current_stream()


def run(parked, steps=20):
    for _ in range(3):
        tr.train_step(audio, lab, audio=True)
    torch.cuda.synchronize()
    base = value()
    t0 = time.perf_counter()
    for s in range(steps):
        if parked:
            assert hip.hipStreamWaitValue32(ctypes.c_void_p(b.cuda_stream), sig, base + s + 1, GTE, 0xFFFFFFFF) == 0
            with torch.cuda.stream(b):
                y.add_(1.0)            # (a kernel with a scalar argument: an H2D copy here would block the host behind the wait)
        tr.train_step(audio, lab, audio=True)
        if parked:
            signal(cur, base + s + 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("(2) ...", flush=True)
import faulthandler
faulthandler.dump_traceback_later(60, exit=True)
print("    alone: %.3f ms" % run(False), flush=True)
print("    parked, 2 steps: %.3f ms" % run(True, 2), flush=True)
for rep in range(2):
    print("(2) 64-segment step: alone %.3f ms   beside a stream parked in hipStreamWaitValue32 for the whole step %.3f ms" % (run(False), run(True)), flush=True)
