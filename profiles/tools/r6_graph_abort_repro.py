"""Round 6: the abort of round 5 (gpurun_out/r5/s21_run2.log) in twenty lines, torch only + the library's status query.

  1. a capture is invalidated (here: a device synchronise inside it — in the long-lived suite process it was the cyclic
     garbage collector freeing an older trainer's graphs / page-locked tables / communicator in the middle of a capture);
  2. CUDAGraph.capture_end() raises; is the stream out of capture mode afterwards?  (printed)
  3. capture_begin() of a NEW CUDAGraph on a stream that is still capturing raises ("Cannot register the state during
     capturing stage") AFTER the graph has noted the default generator's state but BEFORE that state has noted the graph;
  4. destroying that graph object: ~CUDAGraph -> CUDAGeneratorState::unregister_graph -> TORCH_CHECK throws inside a
     destructor -> std::terminate -> SIGABRT ("The graph should be registered to the state").
usage: r6_graph_abort_repro.py [abandon]   (abandon: end the capture through mrmt3_stream_abandon_capture before step 3)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mr-mt3_amd")):
    sys.path.insert(0, p)
import torch

from mrmt3 import lib

lib.abort_trace_install("")
abandon = len(sys.argv) > 1 and sys.argv[1] == "abandon"
x = torch.zeros(16, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    g.capture_begin(capture_error_mode="thread_local")
    x.add_(1.0)
    print("1. capturing:", lib.stream_capture_status(s), flush=True)
    try:
        torch.cuda.synchronize()
    except Exception as e:
        print("   synchronize inside the capture ->", type(e).__name__, str(e).splitlines()[0], flush=True)
    print("   status now:", lib.stream_capture_status(s), "| pending HIP error:", repr(lib.runtime_error_pop()), flush=True)
    try:
        x.add_(1.0)
        print("   a launch after the invalidation: accepted", flush=True)
    except Exception as e:
        print("   a launch after the invalidation ->", str(e).splitlines()[0], flush=True)
    try:
        g.capture_end()
        print("2. capture_end: ok", flush=True)
    except Exception as e:
        print("2. capture_end ->", type(e).__name__, str(e).splitlines()[0], flush=True)
    print("   status after capture_end:", lib.stream_capture_status(s), flush=True)
    if abandon:
        print("   abandon:", lib.stream_abandon_capture(s), "->", lib.stream_capture_status(s), flush=True)
    g2 = torch.cuda.CUDAGraph()
    try:
        g2.capture_begin(capture_error_mode="thread_local")
        print("3. second capture_begin: ok", flush=True)
        x.add_(1.0)
        g2.capture_end()
        g2.replay()
    except Exception as e:
        print("3. second capture_begin ->", type(e).__name__, str(e).splitlines()[0], flush=True)
    print("4. destroying the second graph object ...", flush=True)
    del g2
    import gc
    gc.collect()
    print("   survived", flush=True)
    del g
    gc.collect()
torch.cuda.synchronize()
print("done, x =", x[:2].tolist(), flush=True)
