"""bench.py — MR-MT3 hot path on MI355X: training segments/sec (+ inference real-time factor).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one optimizer step over one batch of synthetic audio per GPU, everything inside the
timed region: log-mel frontend (HIP) -> T5-small encoder/decoder forward (dropout ON) -> fused CE ->
hand-written backward overlapped with the RCCL gradient all-reduce -> one-launch AdamW.
Workload at N=1 = BASELINE.json configs[1]: MT3Net (T5-small) bf16, batch 64 segments of 2.048 s
(32768 samples @16 kHz -> 256 mel frames), 1024-token targets, golden-recipe weights.
Data parallel: every rank draws its own 64 segments (weak scaling); value = N*64*K / max-rank time.

`--gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment) starts the N ranks itself: this process —
which has not touched the GPU — runs `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a
CHILD process, forwards rank 0's JSON line and exits with the child's code (the reference gets its ranks from its launcher
the same way: config/config.yaml:45-46 `devices`, train.sh:6).  More ranks than visible GPUs is refused, and so is a
launcher whose WORLD_SIZE differs from --gpus: the line never reports fewer GPUs than were asked for.

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel family, measured with
events on the launch stream in an instrumented pass of the same steps), `cpu_baseline` (the CPU
oracle timed on this host, N=1 only) and `inference` (greedy decode RTF, hipGraph replayed steps).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "mr-mt3_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
FLOP_PER_SEG_FWD_BWD = 225.1e9   # MT3Net, SURVEY §8d (full causal square counted, as the reference computes it)
FLOP_PER_SEG_MRMT3 = 248.0e9     # segmem_v2_with_prev, 64 memory slots, as written
FLOP_PER_SEG_LONG = 707.5e9      # the same model on 2048-frame segments (BASELINE configs[4])
SEG_SECONDS = 32768 / 16000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="segments per GPU per step")
    ap.add_argument("--variant", default="t5", choices=["t5", "segmem_v2", "segmem_v2_with_prev"])
    ap.add_argument("--mel-frames", type=int, default=256, help="mel frames per segment (2048: BASELINE configs[4])")
    ap.add_argument("--decode-tokens", type=int, default=1024)
    ap.add_argument("--decode-batch", type=int, default=8)
    ap.add_argument("--no-inference", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline", action="store_true",
                    help="with more than one rank: also run the instrumented eager repeat (off by default at N > 1: it is an eager "
                         "pass on EVERY rank; the N = 1 line carries the roofline)")
    ap.add_argument("--no-preflight", action="store_true", help="skip the multi-rank start-up checks")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--extra-batch", type=int, default=12, help="also time this per-GPU batch (0 = off)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip train_mrmt3 / train_mrmt3_b12 / train_long_context (BASELINE configs[2] and [4])")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (the N > 1 code path on one GPU)")
    return ap.parse_args(argv)


def visible_gpus():
    """Devices this process could use, WITHOUT initialising the GPU (device_count() only reads the driver's list): the
    parent of the ranks must stay off the GPU."""
    return int(torch.cuda.device_count())


def launch_ranks(args, argv, launcher=None, n_visible=None, out_fd=1):
    """`bench.py --gpus N` without a launcher around it: run the N ranks as a child `torch.distributed.run`, forward
    rank 0's JSON line to `out_fd`, return the child's exit code.  `launcher` (tests) replaces the command prefix."""
    import socket
    import subprocess
    n_visible = visible_gpus() if n_visible is None else n_visible
    if args.gpus > n_visible:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible: refusing to run fewer ranks than asked for\n"
                         % (args.gpus, n_visible))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child_args = [a for a in argv if a != "--spawn"]
    cmd = (list(launcher) if launcher is not None else
           [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port)]) + [os.path.abspath(__file__)] + child_args
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    env["MRMT3_BENCH_SPAWNED"] = "1"
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err_text = (r.stderr or b"").decode(errors="replace")
    sys.stderr.write(err_text)                               # the ranks' own lines (`rank r: ...`) stay visible to whoever runs this
    sys.stderr.flush()
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        os.write(out_fd, (lines[-1] + "\n").encode())
    elif r.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a JSON line\n")
        return 3
    else:
        # the ranks died without a line of their own (a rank other than 0 failed and the launcher took the rest down, or the
        # process was killed): say where, from the per-rank stage lines, so that the caller still gets ONE JSON line
        os.write(out_fd, (json.dumps(failure_record(args.gpus, err_text, r.returncode)) + "\n").encode())
    return r.returncode


def failure_record(n_gpus, err_text, returncode):
    """The JSON line of a multi-rank run that ended without one: the stage each rank reached last (`rank r: stage NAME ...`
    lines on stderr) and the first failure message."""
    import re
    last, failed = {}, None
    for l in err_text.splitlines():
        m = re.match(r"rank (\d+): (FAILED at stage|stage) ([A-Za-z0-9_]+)(.*)", l)
        if m:
            last[int(m.group(1))] = m.group(3)
            if m.group(2).startswith("FAILED") and failed is None:
                failed = {"rank": int(m.group(1)), "stage": m.group(3), "message": m.group(4).strip(" :")}
    stage = failed["stage"] if failed else (min(last.items(), key=lambda kv: STAGES.index(kv[1]) if kv[1] in STAGES else 99)[1]
                                            if last else "launch")
    return {"metric": METRIC, "value": None, "unit": "segments/s", "n_gpus": n_gpus, "higher_is_better": True,
            "error": (failed["message"] if failed else "the ranks exited with code %d without a result line" % returncode),
            "stage": stage, "failed_rank": failed["rank"] if failed else None,
            "last_stage_per_rank": {str(k): v for k, v in sorted(last.items())}, "returncode": returncode}


METRIC = "train segments/sec (T5-small MT3Net, 256-frame mel, 1024-token target; log-mel + fwd + bwd + AdamW)"
STAGES = ["rccl_init", "model", "bucket_allreduce", "stream_pick", "capture", "warmup", "timed", "roofline", "report"]
_STAGE = {"name": "start", "rank": 0}


def stage(name, msg=""):
    """One short stderr line per rank and stage: the first multi-GPU run is one shot, its tail must say how far each rank got."""
    _STAGE["name"] = name
    sys.stderr.write("rank %d: stage %s%s\n" % (_STAGE["rank"], name, (" " + msg) if msg else ""))
    sys.stderr.flush()


def preflight(trainer, dev, rank, world, step):
    """Start-up of a multi-rank run, BEFORE the timed loop: one all-reduce per gradient bucket size through
    the exact path the step uses, with a checksum; the collective-stream pick; capture of the segmented step + 2 replays and a
    replica check.  Returns the figures for the JSON line; raises on any mismatch."""
    out = {}
    b = trainer.buckets
    G = trainer.flat.G
    rows = []
    for j, bk in enumerate(b.buckets):
        g = G[bk["start"]:bk["end"]]
        g.fill_(float(rank + 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w = b._all_reduce(g, b.collective_stream(dev))
        w.wait()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        want = world * (world + 1) / 2.0
        got = (float(g[0].item()), float(g[-1].item()), float(g.double().mean().item()))
        if any(abs(x - want) > 1e-6 * want for x in got):
            raise RuntimeError("bucket %d (%d elements): all-reduce checksum %r, expected %g" % (j, g.numel(), got, want))
        nbytes = g.numel() * 4
        rows.append({"bucket": j, "MB": nbytes / 1e6, "ms": ms,
                     "busbw_GBps": 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9 if world > 1 else None})
    G.zero_()
    out["bucket_allreduce"] = rows
    stage("bucket_allreduce", "%d buckets ok: %s" % (len(rows), ", ".join("%.0f MB %.2f ms" % (r["MB"], r["ms"]) for r in rows)))
    t0 = time.perf_counter()
    ok = trainer._pick_collective_stream(torch.cuda.current_stream(), dev)
    trainer._collective_stream_checked = True
    out["stream_pick"] = {"side_by_side": bool(ok), "candidates_tried": len(trainer._stream_candidates),
                          "seconds": time.perf_counter() - t0}
    stage("stream_pick", "%s after %d candidate(s), %.2f s" % ("side by side" if ok else "NO stream runs beside the compute stream",
                                                               len(trainer._stream_candidates), out["stream_pick"]["seconds"]))
    t0 = time.perf_counter()
    n = 0
    while trainer.use_graph and not trainer.graph_captured:
        loss = step()
        n += 1
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    if not bool(torch.isfinite(loss).item()):
        raise RuntimeError("loss is not finite after the first steps: %r" % float(loss.item()))
    chk = trainer.flat.P.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    if dist.is_initialized():
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if float(lo.item()) != float(hi.item()):
        raise RuntimeError("the replicas differ after %d steps: weight checksums %r .. %r" % (n + 2, float(lo.item()), float(hi.item())))
    cap = next(iter(trainer._graphs.values())) if trainer._graphs else None
    out["capture"] = {"eager_steps": n, "captured": cap is not None, "graph_segments": (len(cap.segments) + 1) if cap else 0,
                      "seconds": time.perf_counter() - t0, "replicas_identical": True}
    stage("capture", "%s, %d steps, %.2f s, replicas identical, loss %.4f"
          % (("%d graph segments" % out["capture"]["graph_segments"]) if cap else "eager (no graph)", n + 2,
             out["capture"]["seconds"], float(loss.item())))
    return out


def build_model(variant, dev, dtype=torch.bfloat16):
    from mrmt3.synthetic import T5_SMALL
    if variant == "t5":
        from models.t5 import T5ForConditionalGeneration
        m = T5ForConditionalGeneration(T5_SMALL, compute_dtype=dtype)
    elif variant == "segmem_v2":
        from models.t5_segmem_v2 import T5SegMemV2
        m = T5SegMemV2(T5_SMALL, 1, 64, compute_dtype=dtype)
    else:
        from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
        m = T5SegMemV2WithPrev(T5_SMALL, 1, 64, compute_dtype=dtype)
    return m.load_golden().to(dev)


def tn_slab_bytes(batch):
    """Bytes the batched split-K reduction streams (all queued sites' slabs)."""
    return sum(batch._slabs[k].numel() for k in batch._queue) if batch is not None else 0


def roofline_pass(trainer, audio, labels, prev, steps):
    """Instrumented repeat of the timed steps: events around every heavy launch on the launch stream."""
    from mrmt3 import lib
    lib.PROFILE = []
    lib.PROFILE_BYTES.clear()
    for _ in range(steps):
        trainer.train_step(audio, labels, prev, audio=True)
    torch.cuda.synchronize()
    fam = {}
    for name, work, unit, e0, e1 in lib.PROFILE:
        f = fam.setdefault(name, dict(ms=0.0, work=0.0, n=0, unit=unit))
        f["ms"] += e0.elapsed_time(e1)
        f["work"] += work
        f["n"] += 1
    lib.PROFILE = None
    return fam


def timed_workload(dev, variant, B, n_samples, steps, flop_per_seg, lr=2e-4, lr_lambda=None, seed=365):
    """One more training workload beside the headline one, same step (log-mel + fwd + CE + bwd + AdamW, dropout on, graph
    replays): ms per step, segments/s, model TFLOP/s and the three kernel families that take most of the step (events
    around every launch of an eager repeat, as roofline_pass does).  A model and trainer of its own, released at the end."""
    from mrmt3.synthetic import synth_audio, synth_labels
    from mrmt3.trainer import Trainer
    model = build_model(variant, dev)
    tr = Trainer(model, lr=lr, lr_lambda=lr_lambda)
    audio = torch.from_numpy(synth_audio(B, n_samples, seed=seed)).to(dev)
    labels = torch.from_numpy(synth_labels(B, seed=seed)).to(dev)
    prev = torch.from_numpy(synth_labels(B, seed=1000 + seed)).to(dev) if variant == "segmem_v2_with_prev" else None
    step = lambda: tr.train_step(audio, labels, None if prev is None else prev.clone(), audio=True)
    while tr.use_graph and not tr.graph_captured:
        step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"variant": variant, "segments_per_gpu": B, "mel_frames": -(-n_samples // 128), "ms_per_step": 1e3 * dt,
           "segments_per_s": B / dt, "audio_seconds_per_step": B * n_samples / 16000.0,
           "model_tflops": B / dt * flop_per_seg / 1e12, "model_flop_per_segment": flop_per_seg,
           "step_graph": bool(tr.use_graph and tr.graph_captured), "final_loss": float(loss.item()),
           "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}
    graph_was, tr.use_graph = tr.use_graph, False
    was, tr.engine.overlap_wgrad = tr.engine.overlap_wgrad, False
    n_rf = 2
    fam = roofline_pass(tr, audio, labels, None if prev is None else prev.clone(), n_rf)
    tr.engine.overlap_wgrad, tr.use_graph = was, graph_was
    ms = {k: v["ms"] / n_rf for k, v in fam.items() if "@" not in k}
    out["eager_timed_ms_per_step"] = sum(ms.values())
    out["top3_families_ms_per_step"] = dict(sorted(ms.items(), key=lambda kv: -kv[1])[:3])
    del tr, model
    torch.cuda.empty_cache()
    return out


def cpu_baseline(seconds):
    """The CPU oracle (port of the reference path) timed on this host: fwd + CE + backward, B=1."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_audio, synth_labels
    from oracle import logmel_ref, t5_ref
    # torch's intra-op pool stops scaling (and collapses) far below the host's core count on
    # T5-small sized matmuls: use at most 16 threads and report that number as `cores`.
    cores = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL).items()}
    audio = synth_audio(1)
    lab = torch.from_numpy(synth_labels(1))
    n, t0 = 0, time.perf_counter()
    while True:
        mel = torch.from_numpy(logmel_ref.logmel_segments(audio))
        loss = t5_ref.ce_loss(t5_ref.forward_logits(sd, T5_SMALL, mel, lab), lab)
        loss.backward()
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 64:
            break
    dt = time.perf_counter() - t0
    out = dict(value=n / dt, unit="segments/s", cores=torch.get_num_threads(), kind="port",
               sample="%d x (log-mel + fwd + CE + bwd) of 1 segment (no optimizer, no dropout), fp32 torch CPU oracle, %.1f s" % (n, dt))
    # per-stage figures of the same port (BASELINE.md §2 / SURVEY §8d): a few repetitions each, bounded to ~40 s in all
    def best(fn, reps):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return min(ts), sum(ts) / len(ts)
    sd_ng = {k: v.detach() for k, v in sd.items()}
    mel1 = torch.from_numpy(logmel_ref.logmel_segments(audio))
    sd_dec = dict(sd_ng)
    sd_dec["lm_head.weight"] = sd_ng["lm_head.weight"].clone()
    sd_dec["lm_head.weight"][1].zero_()                      # EOS never wins: every requested step runs
    stages = {"threads": cores}
    with torch.no_grad():
        stages["log_mel_ms_per_segment"] = 1e3 * best(lambda: logmel_ref.logmel_segments(audio), 3)[0]
        mn, mean = best(lambda: t5_ref.ce_loss(t5_ref.forward_logits(sd_ng, T5_SMALL, mel1, lab), lab), 3)
        stages["fwd_loss_b1_ms"] = {"min": 1e3 * mn, "mean": 1e3 * mean}
        # the reference's own decode algorithm (no KV cache, full prefix recompute), one segment
        for n_tok in (64, 128, 256):
            t = best(lambda: t5_ref.generate_t5(sd_dec, T5_SMALL, mel1, max_length=n_tok), 1)[0]
            stages["greedy_no_cache_%d_tokens_s" % n_tok] = t
            stages["rtf_no_cache_%d_tokens" % n_tok] = t / SEG_SECONDS
        # the cached algorithm (what the HIP decoder does), all 1024 tokens
        t = best(lambda: t5_ref.generate_t5_cached(sd_dec, T5_SMALL, mel1, max_length=1024), 1)[0]
        stages["greedy_cached_1024_tokens_s"] = t
        stages["rtf_cached_1024_tokens"] = t / SEG_SECONDS

    def fwd_bwd(sdict, melb, labb, **kw):
        loss = t5_ref.ce_loss(t5_ref.forward_logits(sdict, T5_SMALL, melb, labb, **kw), labb)
        loss.backward()
    mel4 = mel1.repeat(4, 1, 1)
    lab4 = torch.from_numpy(synth_labels(4))
    mn, mean = best(lambda: fwd_bwd(sd, mel4, lab4), 2)
    stages["fwd_bwd_b4_segments_per_s"] = {"best": 4.0 / mn, "mean": 4.0 / mean}
    # MR-MT3 proper: segment memory from the previous segment's tokens (64 slots)
    sd_seg = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 1).items()}
    prev1 = torch.from_numpy(synth_labels(1, seed=1365))
    mn, mean = best(lambda: fwd_bwd(sd_seg, mel1, lab, variant="segmem_v2_with_prev", targets_prev=prev1.clone()), 2)
    stages["segmem_v2_with_prev_fwd_bwd_b1_segments_per_s"] = {"best": 1.0 / mn, "mean": 1.0 / mean}
    # the same fwd + CE + bwd at 8 threads (BASELINE.md §2 asks for N = 8 next to the larger pool) ...
    torch.set_num_threads(min(8, cores))
    mn, mean = best(lambda: fwd_bwd(sd, mel1, lab), 3)
    stages["fwd_bwd_b1_8_threads_segments_per_s"] = {"best": 1.0 / mn, "mean": 1.0 / mean}
    torch.set_num_threads(cores)
    # ... and at N = every core of the host, in a child process with a deadline: torch's intra-op pool collapses far
    # below this host's core count on T5-small sized matmuls (round 1: 256 threads took 296 s per step)
    def child_fwd_bwd(threads, deadline):
        """fwd + CE + bwd of one segment at `threads` intra-op threads, in a child process with a deadline."""
        code = ("import sys,time,torch;sys.path[:0]=[%r,%r];from mrmt3.synthetic import T5_SMALL,golden_weights,synth_mel,synth_labels;"
                "from oracle import t5_ref;torch.set_num_threads(%d);sd={k:torch.from_numpy(v).requires_grad_(True) for k,v in golden_weights(T5_SMALL).items()};"
                "mel=torch.from_numpy(synth_mel(1));lab=torch.from_numpy(synth_labels(1));f=lambda:t5_ref.ce_loss(t5_ref.forward_logits(sd,T5_SMALL,mel,lab),lab).backward();"
                "f();ts=[]\nfor _ in range(3):\n t=time.perf_counter();f();ts.append(time.perf_counter()-t)\nprint(min(ts))") % (
                    os.path.join(ROOT, "mr-mt3_amd"), ROOT, threads)
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=deadline)
            return {"threads": threads, "segments_per_s": 1.0 / float(r.stdout.strip().splitlines()[-1])}
        except Exception as e:
            return {"threads": threads, "segments_per_s": None, "note": "no result within %d s (%s)" % (deadline, type(e).__name__)}
    # one thread per PHYSICAL core (BASELINE.md §2 "all physical cores"): distinct (package, core) pairs of /proc/cpuinfo
    try:
        pairs, pkg = set(), None
        for l in open("/proc/cpuinfo"):
            if l.startswith("physical id"):
                pkg = l.split(":")[1].strip()
            elif l.startswith("core id"):
                pairs.add((pkg, l.split(":")[1].strip()))
        phys = len(pairs) or None
    except Exception:
        phys = None
    if phys and phys != cores:
        stages["fwd_bwd_b1_one_thread_per_physical_core"] = child_fwd_bwd(phys, 60)
    ncpu = os.cpu_count() or cores
    if ncpu > cores:
        code = ("import sys,time,torch;sys.path[:0]=[%r,%r];from mrmt3.synthetic import T5_SMALL,golden_weights,synth_mel,synth_labels;"
                "from oracle import t5_ref;torch.set_num_threads(%d);sd={k:torch.from_numpy(v) for k,v in golden_weights(T5_SMALL).items()};"
                "mel=torch.from_numpy(synth_mel(1));lab=torch.from_numpy(synth_labels(1));f=lambda:t5_ref.ce_loss(t5_ref.forward_logits(sd,T5_SMALL,mel,lab),lab);"
                "torch.no_grad().__enter__();f();t=time.perf_counter();f();print(time.perf_counter()-t)") % (
                    os.path.join(ROOT, "mr-mt3_amd"), ROOT, ncpu)
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=45)
            stages["fwd_loss_b1_all_%d_threads_ms" % ncpu] = 1e3 * float(r.stdout.strip().splitlines()[-1])
        except Exception as e:     # deadline passed (or the child failed): say so instead of a number
            stages["fwd_loss_b1_all_%d_threads_ms" % ncpu] = "no result within 45 s (%s)" % type(e).__name__
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        out["cpu_model"] = "unknown"
    out["stages"] = stages
    return out


def inference_rtf(dev, tokens, batch):
    """Greedy decode of `batch` segments x `tokens` tokens (EOS suppressed so every step runs)."""
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio
    m = build_model("t5", dev).eval()
    with torch.no_grad():
        m.flat.master("lm_head.weight")[1].zero_()
    big = 64          # a long recording's segments, or `contiguous_inference`
    huge = 256        # the decoder's maximum group: many recordings' segments decoded together (serving throughput)
    audio = torch.from_numpy(synth_audio(max(batch, huge), seed=366)).to(dev)
    out = {}
    for name, nb in dict([("b1", 1), ("b%d" % batch, batch), ("b%d" % big, big), ("b%d" % huge, huge)]).items():
        a = audio[:nb]
        for rep in range(2):      # first pass captures the graph
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mel = sp.logmel_segments(a, out_bf16=True)
            ids = m.generate(mel, max_length=tokens)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert ids.shape == (nb, tokens + 1)
        # bytes one decode step has to move: 45.6 MB of bf16 decoder weights + lm_head + this step's K/V:
        # the self-attention cache (mean length tokens/2) and the projected encoder states (256 frames), 8 layers x
        # 2 x 384 x 2 B per position
        # ALGORITHMIC bytes of a step: the weights once + this step's K/V.  What the kernels actually stream is more —
        # they re-read the weights once per sequence (<= 8 sequences) or per group of 16 — and is reported next to it
        # as wasted traffic, not counted as achieved bandwidth.
        w_bytes = 45.6e6
        rereads = nb if nb <= 8 else -(-nb // 16)
        kv_bytes = nb * 8 * 2 * 384 * 2 * (tokens / 2 + 256)
        step_s = dt / tokens
        out[name] = dict(segments=nb, tokens=tokens, seconds=dt, rtf=dt / (nb * SEG_SECONDS),
                         ms_per_token_step=1e3 * step_s, graph=bool(m._decoder.graph_captured),
                         roofline_decode={"bound": "hbm", "bytes_per_step": w_bytes + kv_bytes,
                                          "achieved": (w_bytes + kv_bytes) / step_s / 1e9, "peak": PEAK_HBM_GBS,
                                          "unit": "GB/s", "frac": (w_bytes + kv_bytes) / step_s / 1e9 / PEAK_HBM_GBS,
                                          "weight_rereads_per_step": rereads,
                                          "streamed_bytes_per_step": w_bytes * rereads + kv_bytes,
                                          "launch_chain_floor_ms": 66 * 1.77e-3})
    # MR-MT3 proper (segment memory from the previous segment's tokens): a recording is a sequential chain,
    # several recordings decode in lockstep, one batch row each
    del m
    mm = build_model("segmem_v2_with_prev", dev).eval()
    with torch.no_grad():
        mm.flat.master("lm_head.weight")[1].zero_()
    n_seg = 3
    for name, n_songs in (("mrmt3_1song", 1), ("mrmt3_8songs", 8), ("mrmt3_64songs", 64)):
        songs = [sp.logmel_segments(audio[i * n_seg:(i + 1) * n_seg], out_bf16=True) for i in range(n_songs)]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids = mm.generate_songs(songs, max_length=tokens)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert len(ids) == n_songs and ids[0].shape == (n_seg, tokens)
        out[name] = dict(recordings=n_songs, segments_each=n_seg, tokens=tokens, seconds=dt,
                         rtf=dt / (n_songs * n_seg * SEG_SECONDS))
    return out


PMC_TABLES = ("r06_pmc_step_traffic.json", "r05_pmc_step_traffic.json", "r04_pmc_step_traffic.json", "r03_pmc_step_traffic.json")


def pmc_table(batch, lib_version):
    """The newest committed whole-step PMC table (bytes per family and per step) that was measured on this library
    version and at this batch size: (table, name, None), else (None, None, reason).  Tables carry `lib_version` and
    `batch` stamps (r03's has none: it is version 104 at 64 segments); a stale table must not ride along in a bench
    line."""
    why = []
    for name in PMC_TABLES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as fh:
                tab = json.load(fh)
        except Exception as e:
            why.append("%s unreadable (%s)" % (name, e))
            continue
        tv, tb = int(tab.get("lib_version", 104)), int(tab.get("batch", 64))
        if tv != lib_version:
            why.append("%s was measured on library version %d, this is %d" % (name, tv, lib_version))
        elif tb != batch:
            why.append("%s is a %d-segment table, this run has %d" % (name, tb, batch))
        else:
            return tab, "profiles/" + name, None
    return None, None, "no PMC table for this run: " + ("; ".join(why) if why else "none committed")


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(launch_ranks(args, argv))                 # (nothing above has touched the GPU)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d under a launcher with WORLD_SIZE=%s: they must agree\n"
                         % (args.gpus, os.environ.get("WORLD_SIZE")))
        sys.exit(2)
    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 while the bench
    # runs (RCCL prints a five-line version banner there when its first communicator is created) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _STAGE["rank"] = rank
    try:
        run(args, world, rank, local, json_fd)
    except BaseException as e:     # noqa: BLE001 — whatever stops a rank: its stage on stderr, rank 0's JSON line with "error"
        if isinstance(e, SystemExit) and not e.code:
            raise
        import traceback
        traceback.print_exc()
        msg = "%s: %s" % (type(e).__name__, (str(e).splitlines() or [""])[0])
        sys.stderr.write("rank %d: FAILED at stage %s: %s\n" % (rank, _STAGE["name"], msg))
        sys.stderr.flush()
        if rank == 0:
            os.write(json_fd, (json.dumps({"metric": METRIC, "value": None, "unit": "segments/s", "n_gpus": world,
                                           "higher_is_better": True, "error": msg, "stage": _STAGE["name"],
                                           "failed_rank": rank}) + "\n").encode())
        os._exit(1)                # no teardown through a half-initialised process group: the launcher ends the other ranks


def run(args, world, rank, local, json_fd):
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no CPU fallback"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_coll = os.environ.get("MRMT3_DDP_FORCE_COLLECTIVES") == "1" and "MASTER_ADDR" in os.environ
    multi = world > 1 or force_coll
    t_init = None
    if multi:                      # (force: the RCCL bucket path at world size 1, for the record in profiles/)
        stage("rccl_init", "world %d, device %d, HSA_ENABLE_IPC_MODE_LEGACY=%s" % (world, local, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
        t0 = time.perf_counter()
        dist.init_process_group("nccl", device_id=dev)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)       # the communicator is made here, not in init_process_group
        torch.cuda.synchronize()
        t_init = time.perf_counter() - t0
        if float(one.item()) != float(world):
            raise RuntimeError("first all-reduce: sum of ones over %d ranks came back as %r" % (world, float(one.item())))
        stage("rccl_init", "done in %.2f s" % t_init)
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1      # what RCCL's communicator says, not the flag
    assert ranks_seen == world, (ranks_seen, world)
    from mrmt3 import lib
    lib.load()
    from mrmt3.synthetic import synth_audio, synth_labels
    from mrmt3.trainer import Trainer
    from utils import cosine_warmup_lambda

    B = args.batch
    if multi:
        stage("model")
    model = build_model(args.variant, dev)
    trainer = Trainer(model, lr=2e-4, lr_lambda=cosine_warmup_lambda(64500, 1289 * 800, min_lr=1e-4))
    n_samples = args.mel_frames * 128
    flop_per_seg = ({"t5": FLOP_PER_SEG_FWD_BWD}.get(args.variant, FLOP_PER_SEG_MRMT3) if args.mel_frames == 256 else
                    FLOP_PER_SEG_LONG if (args.mel_frames == 2048 and args.variant == "segmem_v2_with_prev") else None)
    audio = torch.from_numpy(synth_audio(B, n_samples, seed=365 + rank)).to(dev)
    labels = torch.from_numpy(synth_labels(B, seed=365 + rank)).to(dev)
    prev = torch.from_numpy(synth_labels(B, seed=1365 + rank)).to(dev) if args.variant == "segmem_v2_with_prev" else None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not measured: the trainer runs its first steps eagerly and then captures the step into hipGraphs
    # (mrmt3/trainer.py); make sure that has happened before the W warm-up steps, whatever W is
    pre = None
    if multi and not args.no_preflight:
        pre = preflight(trainer, dev, rank, world,
                        lambda: trainer.train_step(audio, labels, None if prev is None else prev.clone(), audio=True))
        pre["rccl_init_seconds"] = t_init
    while trainer.use_graph and not trainer.graph_captured:
        trainer.train_step(audio, labels, None if prev is None else prev.clone(), audio=True)
    if multi:
        stage("warmup")
    for _ in range(args.warmup):
        loss = trainer.train_step(audio, labels, None if prev is None else prev.clone(), audio=True)
    sync()
    if multi:
        stage("timed")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(audio, labels, None if prev is None else prev.clone(), audio=True)
    host_issue = time.perf_counter() - t0          # host time to enqueue the steps (device still running)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    seg_per_s = world * B * args.steps / dt

    if multi:
        stage("timed", "%.2f ms per step" % (1e3 * dt / args.steps))
    res = {
        "metric": METRIC,
        "value": seg_per_s, "unit": "segments/s", "n_gpus": world, "ranks_seen": ranks_seen,
        "launched_by": ("bench.py --gpus %d -> child torch.distributed.run" % args.gpus if os.environ.get("MRMT3_BENCH_SPAWNED")
                        else "external launcher (WORLD_SIZE=%d)" % world if "WORLD_SIZE" in os.environ else "single process"),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%s]: %s bf16, %d segments/GPU/step of %d samples @16kHz -> %dx512 mel, "
                               "1024-token targets, dropout 0.1 on, golden-recipe weights"
                               % ("1" if args.variant == "t5" else "4" if args.mel_frames == 2048 else "2", args.variant, B,
                                  n_samples, args.mel_frames),
                   "segments_per_gpu": B, "global_segments": B * world, "parallelism": "dp%d" % world,
                   "audio_seconds_per_step": B * world * n_samples / 16000.0},
        "final_loss": final_loss,
        "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
        "step_graph": bool(trainer.use_graph and trainer.graph_captured),
        # graphs replayed per step: the compute step (one per gradient bucket: the collectives run eagerly between them) + AdamW's tail
        "graph_segments": (len(next(iter(trainer._graphs.values())).segments) + 1) if trainer._graphs else 0,
        "collectives": ("rccl%s, %d buckets per step%s" % (" through the C ABI (mrmt3_allreduce)" if trainer.buckets.native else " through torch.distributed",
                                                           len(trainer.buckets.buckets), " (forced at world 1)" if force_coll and world == 1 else "")
                        if trainer.buckets.active else "none (world 1)"),
        # the grouped weight-gradient launch of the last join (with collectives: of the last gradient bucket)
        "weight_gradient_launch": (lambda g: None if g is None or g.last_info is None else
                                   {"items": int(g.last_info.n_items), "gradient_tiles": int(g.last_info.n_rtiles),
                                    "items_per_workgroup_max": int(g.last_info.rounds), "workgroups": int(g.last_info.n_ctas),
                                    "partial_tile_bytes": int(g.last_info.slab_bytes)})(trainer.engine.tn_group),
        "model_tflops": None if flop_per_seg is None else seg_per_s * flop_per_seg / 1e12 / world,
        "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
    }
    if pre is not None:
        res["preflight"] = pre
    # the instrumented eager repeat: at N = 1 by default; at N > 1 only with --roofline (it is an eager pass on every rank)
    do_roofline = not args.no_roofline and (world == 1 or args.roofline)
    if rank == 0 and world == 1 and args.extra_batch > 0:
        # the reference's own per-GPU batch (config_slakh_segmem.yaml: num_rows_per_batch 12)
        Bx = args.extra_batch
        ax, lx = audio[:Bx].contiguous(), labels[:Bx].contiguous()
        px = None if prev is None else prev[:Bx].contiguous()
        for _ in range(4):
            trainer.train_step(ax, lx, None if px is None else px.clone(), audio=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_step(ax, lx, None if px is None else px.clone(), audio=True)
        torch.cuda.synchronize()
        dtx = time.perf_counter() - t1
        res["train_b%d" % Bx] = {"segments_per_gpu": Bx, "ms_per_step": 1e3 * dtx / args.steps,
                                 "segments_per_s": Bx * args.steps / dtx,
                                 "model_tflops": None if flop_per_seg is None else Bx * args.steps / dtx * flop_per_seg / 1e12}
    if do_roofline:
        if multi:
            stage("roofline")
        # every rank repeats the steps (the gradient exchange is collective); rank 0 keeps the timings.
        # The replayed graph is ONE chain of kernels (the grouped weight-gradient launch included), so the eager repeat
        # runs everything on one stream too and every launch is timed the way the graph executes it.
        n_rf = max(1, min(args.steps, 3))
        graph_was, trainer.use_graph = trainer.use_graph, False      # per-launch events need eager launches
        eng = trainer.engine
        was, eng.overlap_wgrad = eng.overlap_wgrad, False
        fam = roofline_pass(trainer, audio, labels, None if prev is None else prev.clone(), n_rf)
        eng.overlap_wgrad = was
        trainer.use_graph = graph_was
    if rank == 0 and do_roofline:
        dom = "gemm_nt_bf16"        # forward + dgrad Linear layers: the family with the most FLOPs per step

        def rate(v):
            return v["work"] / (v["ms"] * 1e-3) / (1e12 if v["unit"] == "FLOP" else 1e9)

        f = fam[dom]
        # HBM traffic of the dominant family: rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction
        # of MI355X_MICROARCH.md) cannot run inside this process; the whole-step table they produced for this tree is
        # committed (profiles/r0N_pmc_step_traffic.json, made by profiles/tools/pmc_step_traffic.sh) and folded in here —
        # only when the table was measured on THIS library version and at THIS batch (pmc_table() says why not otherwise).
        from mrmt3 import lib as _lib
        tab, tab_name, tab_note = pmc_table(B, _lib.load().mrmt3_version())
        traffic = None
        if tab is not None and "gemm_nt" in tab.get("families", {}):
            # measured: the family's row of the whole-step PMC table (2 x FETCH_SIZE + WRITE_SIZE over one eager step,
            # separate passes); algorithmic: operands once + output once, summed over this run's launches of the family
            fam_tab = tab["families"]["gemm_nt"]
            meas = (fam_tab["read_bytes"] + fam_tab["write_bytes"]) / fam_tab["launches"]
            alg = _lib.PROFILE_BYTES.get(dom, 0.0) / max(f["n"], 1)
            traffic = {"bytes_per_launch": meas, "algorithmic_bytes_per_launch": alg, "ratio": meas / alg if alg else None,
                       "source": "%s (%d launches of the family per step incl. the lm_head chunks launched inside "
                                 "mrmt3_lmhead_ce_fwd_bwd)" % (tab_name, fam_tab["launches"])}
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": rate(f), "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": rate(f) / PEAK_BF16_TFLOPS, "traffic": traffic,
                           "launches": f["n"], "avg_launch_ms": f["ms"] / f["n"],
                           "note": "HIP events around every launch of the family in an eager repeat of the timed steps, all "
                                   "kernels on one stream exactly as the replayed graph runs them (the timed steps "
                                   "themselves are graph replays: events cannot be placed inside); the 16 wi projections "
                                   "(GEMM + gated-GELU epilogue in one launch) are their own family gemm_nt_geglu_bf16, "
                                   "priced at the GEMM's FLOPs; gemm_tn_bf16 = the grouped weight-gradient launch + its reduce.  "
                                   "What bounds the family (round 6, profiles/r06_gemm_hidden_stores.txt): a 256 x 256 x 512 tile moves "
                                   "512 KB in and 128 KB out through its CU's vector-memory path in ~15 us (additive) against 8 us of "
                                   "MFMA: these short-K bf16-output products sit at ~0.35 of the MFMA peak for that reason, with a "
                                   "ping-pong schedule and with a one-wave-per-SIMD schedule with trickled stores alike",
                           "families_ms_per_step": {k: v["ms"] / n_rf for k, v in fam.items() if "@" not in k},
                           # attn_*: priced at the FULL score square, as the reference computes it;
                           # attn_*@executed: the FLOPs the kernels issue (causal key tiles above the diagonal skipped)
                           "families_achieved": {k: rate(v) for k, v in fam.items()}}
        # HBM side of the whole step: bytes per step from the committed whole-step PMC table (profiles/tools/
        # pmc_step_traffic.sh) over this run's step time, next to the MFMA fraction (the step holds 14.4 TFLOP)
        step_s = dt / args.steps
        fps = flop_per_seg or 0.0
        res["roofline"]["step"] = {"mfma_TFLOPs": B * fps / step_s / 1e12,
                                   "mfma_frac": B * fps / step_s / 1e12 / PEAK_BF16_TFLOPS}
        if tab is not None:
            res["roofline"]["step"].update({"step_bytes": tab["step_bytes"], "hbm_GBps": tab["step_bytes"] / step_s / 1e9,
                                            "hbm_frac": tab["step_bytes"] / step_s / 1e9 / PEAK_HBM_GBS, "source": tab_name})
        else:
            res["roofline"]["step"].update({"step_bytes": None, "hbm_GBps": None, "hbm_frac": None, "source": tab_note})
        if traffic is None:
            res["roofline"]["traffic_note"] = tab_note
    if rank == 0 and world == 1 and not args.no_extra_workloads:
        # MR-MT3's own model (BASELINE configs[2]: config_slakh_segmem.yaml, segment memory from the previous segment's
        # tokens, models/t5_segmem_v2_with_prev.py:118-128) at the benchmark's 64 segments per GPU and at the reference's
        # 12 (num_rows_per_batch), and configs[4]: the finetune config (bare AdamW, lr 1e-5) on 2048-frame segments + 64
        # memory slots, 12 segments of 16.4 s per GPU.  Each is the same timed step as the headline line.
        lam = cosine_warmup_lambda(64500, 1289 * 800, min_lr=1e-4)
        res["train_mrmt3"] = timed_workload(dev, "segmem_v2_with_prev", B, 32768, args.steps, FLOP_PER_SEG_MRMT3, lr_lambda=lam)
        res["train_mrmt3_b12"] = timed_workload(dev, "segmem_v2_with_prev", 12, 32768, args.steps, FLOP_PER_SEG_MRMT3, lr_lambda=lam)
        res["train_long_context"] = timed_workload(dev, "segmem_v2_with_prev", 12, 2048 * 128, args.steps, FLOP_PER_SEG_LONG,
                                                   lr=1e-5)
    sync()
    trainer.close()                                # graphs first, then the library's own communicator if one was made (MRMT3_DDP_NATIVE)
    if rank == 0 and not args.no_inference:
        del trainer, model
        torch.cuda.empty_cache()
        res["inference"] = inference_rtf(dev, args.decode_tokens, args.decode_batch)
        res["inference"]["metric"] = "real-time factor = wall seconds / audio seconds (greedy, bf16, KV cache, hipGraph)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    sync()
    if multi:
        stage("report")
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if world > 1 or force_coll:
        dist.destroy_process_group()
    # orderly teardown while the interpreter and the HIP runtime are still whole (captured graphs, page-locked tables): the line
    # is out; nothing after it should be able to turn the exit code
    import gc
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
