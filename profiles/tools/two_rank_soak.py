"""Hunt for the two-rank mismatch of DESIGN §6: the 5-step eager training run of tests/test_ddp_gpu.py,
repeated N times inside long-lived processes, every repetition's gradients and weights compared bit for bit with the
first repetition's.

  mode ddp   : two ranks on cuda:0 exchanging over gloo (the test's configuration)
  mode solo2 : two INDEPENDENT processes (no process group) time-slicing cuda:0 — separates gloo from the kernels
  mode solo1 : one process alone (control)

    python3 profiles/tools/two_rank_soak.py <mode> <iterations> [graph]

Environment switches of the engine (MRMT3_WGRAD_STREAM=0, MRMT3_TN_BATCH=0, MRMT3_NORM_DW_BATCH=0, MRMT3_TN_GROUP=0) are
inherited by the workers: a mismatch is bisected by repeating the run with one of them set."""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, mode, iters, port, graph):
    sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
    import torch
    import torch.distributed as dist
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_audio, synth_labels
    from mrmt3.trainer import Trainer
    from models.t5 import T5ForConditionalGeneration
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if mode == "ddp":
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
        dist.init_process_group("gloo", rank=rank, world_size=2)
    w = golden_weights(T5_SMALL)
    audio = torch.from_numpy(synth_audio(2, seed=50 + rank)).to(dev)
    lab = torch.from_numpy(synth_labels(2, 128, seed=60 + rank)).to(dev)
    first = None
    bad = 0
    t0 = time.time()
    for it in range(iters):
        m = T5ForConditionalGeneration(dict(T5_SMALL, dropout_rate=0.0))
        with torch.no_grad():
            m.flat.load_numpy(w)
        m = m.to(dev)
        if rank == 1 and mode == "ddp":
            with torch.no_grad():
                m.flat.P.mul_(1.5)
        tr = Trainer(m, lr=1e-3, graph=graph)
        for _ in range(5):
            tr.train_step(audio, lab, audio=True)
        torch.cuda.synchronize()
        cur = (m.flat.G.clone(), m.flat.P.clone())
        if first is None:
            first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
            bad += 1
            rep = []
            for key in m.flat.shapes:
                a, b = m.flat.view(cur[0], key), m.flat.view(first[0], key)
                n = int((a != b).sum().item())
                if n:
                    rep.append("%s: %d elements, max|d| %.3e" % (key, n, (a - b).abs().max().item()))
            print("rank %d iteration %d DIFFERS from iteration 0: %s" % (rank, it, "; ".join(rep[:12]) +
                                                                        (" ... (%d tensors)" % len(rep) if len(rep) > 12 else "")), flush=True)
        del tr, m
    print("rank %d mode %s graph %d: %d iterations, %d differing from the first, %.1f s" % (rank, mode, graph, iters, bad, time.time() - t0),
          flush=True)
    if mode == "ddp":
        dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
        sys.exit(0)
    mode, iters = sys.argv[1], int(sys.argv[2])
    graph = int(len(sys.argv) > 3 and sys.argv[3] == "graph")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = 1 if mode == "solo1" else 2
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), mode, str(iters), str(port), str(graph)])
             for r in range(n)]
    rc = [p.wait() for p in procs]
    print("exit codes", rc)
    sys.exit(max(rc))
