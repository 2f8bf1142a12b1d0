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
    # MRMT3_SOAK_TRACE=1: an exact (integer, order-independent) checksum of the output of every kernel wrapper of
    # mrmt3.lib, of the weights, their bf16 shadows and transposed copies after every step: the FIRST checksum that
    # differs from the first repetition's names the kernel that produced different bits
    trace = os.environ.get("MRMT3_SOAK_TRACE") == "1"
    log = []
    if trace:
        from mrmt3 import lib as L

        def bits(t):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            v = t.view(torch.int16) if t.element_size() == 2 else (t.view(torch.int32) if t.element_size() == 4 else t.view(torch.int64))
            return v.sum(dtype=torch.int64)

        def wrap(name):
            fn = getattr(L, name)

            def inner(*a, **kw):
                out = fn(*a, **kw)
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for j, o in enumerate(outs):
                    if name in ("lmhead_cross_entropy", "cross_entropy") and j == 0:
                        continue               # the logged loss: the one float-atomic sum of the step (order-dependent last bits)
                    if isinstance(o, torch.Tensor) and o.is_cuda:
                        log.append(("%s[%d] %s" % (name, j, tuple(o.shape)), bits(o)))
                return out
            setattr(L, name, inner)
        for nm in ("logmel", "gemm_nt", "add_rmsnorm_fwd", "attn_fwd", "geglu_fwd", "gemm_nt_geglu", "embed_fwd", "addpos_fwd",
                   "lmhead_cross_entropy", "cross_entropy", "add_rmsnorm_bwd", "attn_bwd", "geglu_bwd", "dropmask_cast", "cast"):
            wrap(nm)
    w = golden_weights(T5_SMALL)
    audio = torch.from_numpy(synth_audio(2, seed=50 + rank)).to(dev)
    use_mel = os.environ.get("MRMT3_SOAK_MEL") == "1"      # feed the mel (made once, before the loop) instead of audio
    if use_mel:
        from contrib import spectrograms as sp
        mels = [sp.logmel_segments(audio, out_bf16=True) for _ in range(3)]
        torch.cuda.synchronize()
        assert torch.equal(mels[0], mels[1]) and torch.equal(mels[1], mels[2]), "three log-mel launches of the same audio differ"
        audio = mels[0]
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
        cur = []
        del log[:]
        for st_ in range(5):
            loss = tr.train_step(audio, lab, audio=not use_mel)
            cur.append((m.flat.G.clone(), loss.clone()))         # the gradient of every step (before AdamW spreads a glitch)
            if trace:
                log.append(("step %d: G" % st_, bits(m.flat.G)))
                log.append(("step %d: P after AdamW" % st_, bits(m.flat.P)))
                log.append(("step %d: bf16 shadow" % st_, bits(m.flat.S)))
                log.append(("step %d: transposed shadow" % st_, bits(m.flat.ST)))
                log.append(("step %d: AdamW m" % st_, bits(m.flat.M)))
                log.append(("step %d: AdamW v" % st_, bits(m.flat.V)))
        torch.cuda.synchronize()
        if trace:
            names = [n for n, _ in log]
            vec = torch.stack([v for _, v in log]).cpu()
            if first is None:
                first_vec, first_names = vec, names
            elif names != first_names or not torch.equal(vec, first_vec):
                j = next((i for i in range(min(len(names), len(first_names))) if names[i] != first_names[i] or vec[i] != first_vec[i]), -1)
                print("rank %d iteration %d: first differing checksum is #%d of %d: %s  (previous ops: %s)" % (
                    rank, it, j, len(names), names[j] if j >= 0 else "?", " <- ".join(names[max(0, j - 3):j][::-1])), flush=True)
        if first is None:
            first = cur
        else:
            for st, ((g, l), (g0, l0)) in enumerate(zip(cur, first)):
                if torch.equal(g, g0):
                    continue
                bad += 1
                same, diff = [], []
                for key in m.flat.shapes:
                    a, b = m.flat.view(g, key), m.flat.view(g0, key)
                    n = int((a != b).sum().item())
                    (diff if n else same).append((key, n, (a - b).abs().max().item() if n else 0.0))
                short = lambda k: k.replace(".weight", "").replace("block.", "").replace("layer.", "").replace("SelfAttention", "sa").replace("EncDecAttention", "ca").replace("DenseReluDense", "ff")
                print("rank %d iteration %d: gradient of step %d DIFFERS (loss %.7f vs %.7f); %d tensors differ, %d identical.\n"
                      "   differing: %s\n   identical: %s" % (
                          rank, it, st, l.item(), l0.item(), len(diff), len(same),
                          "; ".join("%s %d (%.1e)" % (short(k), n, d) for k, n, d in diff[:200]),
                          ", ".join(short(k) for k, _, _ in same[:200])), flush=True)
                break
        del tr, m, cur
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
