"""Is the log-mel kernel bitwise repeatable when two processes share the GPU?  (The two-process soak of
profiles/tools/two_rank_soak.py names it as the first kernel whose output differs.)  Every process transforms the same
audio N times and compares each result with its first; a difference is printed element by element.
    python3 profiles/tools/logmel_repeat.py <processes> <iterations> [segments = 2] [bf16 = 1]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, iters, B, bf16):
    sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
    import torch
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio
    dev = torch.device("cuda:0")
    audio = torch.from_numpy(synth_audio(B, seed=50 + rank)).to(dev)
    noise = torch.randn(2048, 2048, device=dev)
    ref = sp.logmel_segments(audio, out_bf16=bool(bf16)).clone()
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    for it in range(iters):
        if it % 3 == 0:
            noise = noise @ noise * 1e-3          # unrelated work in between: different timing every time
        out = sp.logmel_segments(audio, out_bf16=bool(bf16))
        if not torch.equal(out, ref):
            bad += 1
            d = (out != ref).nonzero()
            frames = sorted({(int(a), int(b)) for a, b, _ in d.tolist()})
            print("rank %d iteration %d: %d elements differ in %d (segment, frame) rows: %s; first: ref %r got %r at %s" % (
                rank, it, d.shape[0], len(frames), frames[:8], float(ref[tuple(d[0])]), float(out[tuple(d[0])]), d[0].tolist()), flush=True)
    print("rank %d: %d iterations of logmel on %d segments (%s out), %d differing from the first, %.1f s" % (
        rank, iters, B, "bf16" if bf16 else "f32", bad, time.time() - t0), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(*[int(a) for a in sys.argv[2:6]])
        sys.exit(0)
    n, iters = int(sys.argv[1]), int(sys.argv[2])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    bf16 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(iters), str(B), str(bf16)]) for r in range(n)]
    sys.exit(max(p.wait() for p in procs))
