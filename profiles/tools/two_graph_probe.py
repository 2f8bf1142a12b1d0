"""Do two hipGraphs replayed on two streams run SIDE BY SIDE on this runtime (ROCm 7.2, MI355X)?  The data-parallel step with
its collectives captured (MRMT3_DDP_GRAPH=1) is a compute graph and a collective graph that hand over through spinning flag
waits (mrmt3_flag_wait): if the runtime puts both graphs behind one another — one hardware queue for both streams, or graph
launches serialised — the first wait spins until its timeout.  Cases: eager kernels / graphs; the producer on the default (null)
stream or on a stream of its own; the consumer stream at normal or at high priority; with N extra streams created first
(HIP multiplexes streams of one priority onto GPU_MAX_HW_QUEUES = 4 hardware queues).
    python3 profiles/tools/two_graph_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
import torch
from mrmt3 import lib

dev = torch.device("cuda:0")
lib.load()
x = torch.zeros(1 << 22, device=dev)
y = torch.zeros(1, device=dev)
TIMEOUT_MS = 500


def case(name, graphs, producer_stream, consumer_stream, rounds=4):
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    seen = torch.zeros(1, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def consume():
        lib.flag_wait(flag, seen, err, TIMEOUT_MS, stream=consumer_stream)
        y.copy_(x[-1:], non_blocking=True)

    def produce():
        for _ in range(8):
            x.add_(1.0)
        lib.flag_signal(flag, stream=producer_stream)

    if graphs:
        gb, ga = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(consumer_stream):
            gb.capture_begin(capture_error_mode="thread_local")
            consume()
            gb.capture_end()
        cap.wait_stream(torch.cuda.current_stream())
        ps_cap = producer_stream if producer_stream.cuda_stream != 0 else cap      # (capture is illegal on the null stream)
        saved = producer_stream
        with torch.cuda.stream(ps_cap):
            ga.capture_begin(capture_error_mode="thread_local")
            for _ in range(8):
                x.add_(1.0)
            lib.flag_signal(flag, stream=ps_cap)
            ga.capture_end()
        torch.cuda.synchronize()
    t = []
    for r in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(consumer_stream):          # the consumer first, like the trainer's collective graph
            gb.replay() if graphs else consume()
        with torch.cuda.stream(producer_stream):
            ga.replay() if graphs else produce()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    ok = int(err.item()) == 0
    print("  %-86s %s   ms per round: %s" % (name, "side by side" if ok else "SERIALISED (wait timed out)",
                                            " ".join("%.1f" % v for v in t)), flush=True)
    return ok


print("two streams, a spinning wait on one, its signal on the other (timeout %d ms); torch %s" % (TIMEOUT_MS, torch.__version__))
null = torch.cuda.default_stream()
for extra in (0, 6):
    keep = [torch.cuda.Stream() for _ in range(extra)]
    for s in keep:
        with torch.cuda.stream(s):
            y.add_(0.0)
    torch.cuda.synchronize()
    tag = " [%d other streams in use]" % extra if extra else ""
    for graphs in (False, True):
        kind = "two GRAPHS" if graphs else "eager kernels"
        case(kind + ": producer on the default stream, consumer normal priority" + tag, graphs, null, torch.cuda.Stream())
        case(kind + ": producer on the default stream, consumer HIGH priority" + tag, graphs, null, torch.cuda.Stream(priority=-1))
        case(kind + ": producer on its own stream, consumer normal priority" + tag, graphs, torch.cuda.Stream(), torch.cuda.Stream())
        case(kind + ": producer on its own stream, consumer HIGH priority" + tag, graphs, torch.cuda.Stream(), torch.cuda.Stream(priority=-1))
