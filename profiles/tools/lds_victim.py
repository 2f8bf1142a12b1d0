"""Which kernel disturbs a co-resident workgroup of ANOTHER process?  The log-mel kernel (24 KB of plain LDS per
workgroup) is bitwise repeatable alone and beside a second log-mel loop, but not beside a training process
(profiles/r03_two_process_soak.txt).  Here a victim process loops log-mel and compares every result with its first,
while an aggressor process loops ONE kernel family at benchmark shapes; one victim / aggressor pair per family.
    python3 profiles/tools/lds_victim.py [seconds per family = 20]
    python3 profiles/tools/lds_victim.py --inprocess [seconds per family = 20] [families]
--inprocess (round 4): ONE process, the log-mel loop on stream A and the aggressor family on stream B
at the same time — the product's own configuration whenever the eager two-stream path or RCCL kernels run beside
compute.  Every victim launch is compared with the first result on the device."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAMILIES = ["gemm_nt8", "gemm_nt_tile", "gemm_nt_geglu", "tn_group", "gemm_tn_tile", "attn_fwd", "attn_bwd", "attn_bwd_onepass",
            "rowops", "ce_adamw", "torch_matmul"]


def aggressor(family, seconds, build_only=False):
    sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
    import torch
    from mrmt3 import lib
    dev = torch.device("cuda:0")
    M = 65536
    bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
    if family == "gemm_nt8":
        a, b = bf(M, 512), bf(512, 512)
        fn = lambda: lib.gemm_nt(a, b)
    elif family == "gemm_nt_tile":
        os.environ["MRMT3_GEMM8"] = "0"
        a, b = bf(M, 512), bf(1152, 512)
        a2, b2 = bf(512, 512), bf(384, 512)
        fn = lambda: (lib.gemm_nt(a, b), lib.gemm_nt(a2, b2), lib.gemm_nt(a.float()[:4096], b.float()))
    elif family == "gemm_nt_geglu":
        a, b = bf(M, 512), bf(2048, 512)
        fn = lambda: lib.gemm_nt_geglu(a, b, p=0.1, seed=1, stream_id=2)
    elif family == "tn_group":
        g = lib.TnGroup()
        ops = [(bf(M, 512), bf(M, 512), torch.zeros(512, 512, device=dev)) for _ in range(6)]
        def fn():
            for x, y, o in ops:
                g.add(x, y, o)
            g.flush()
    elif family == "gemm_tn_tile":
        a, b, o = bf(512, 384), bf(512, 512), torch.zeros(384, 512, device=dev)
        a2, b2, o2 = bf(M, 512), bf(M, 384), torch.zeros(512, 384, device=dev)
        os.environ["MRMT3_TN8"] = "0"
        fn = lambda: (lib.gemm_tn(a, b, o), lib.gemm_tn(a2, b2, o2))
    elif family in ("attn_fwd", "attn_bwd", "attn_bwd_onepass"):
        B, H, L = 64, 6, 1024
        Lk = 256 if family == "attn_bwd_onepass" else L
        causal = family != "attn_bwd_onepass"
        q, k, v = bf(B * L, 384), bf(B * Lk, 384), bf(B * Lk, 384)
        o, lse = lib.attn_fwd(q, k, v, B, H, L, Lk, causal, p=0.1, seed=1, stream_id=1)
        d_o, dq, dk, dv = bf(B * L, 384), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        if family == "attn_fwd":
            fn = lambda: lib.attn_fwd(q, k, v, B, H, L, Lk, causal, p=0.1, seed=1, stream_id=1)
        else:
            fn = lambda: lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, Lk, causal, p=0.1, seed=1, stream_id=1)
    elif family == "rowops":
        x, y, w = torch.randn(M, 512, device=dev), bf(M, 512), torch.ones(512, device=dev)
        dw = torch.zeros(512, device=dev)
        h, dg = bf(M, 2048), bf(M, 1024)
        def fn():
            x1, xn, rstd = lib.add_rmsnorm_fwd(x, y, w, 1e-6, torch.bfloat16, p=0.1, seed=1, stream_y=3)
            lib.add_rmsnorm_bwd(y, None, x1, rstd, w, dw, p=0.1, seed=1, stream_y=3)
            lib.geglu_bwd(h, dg, p=0.1, seed=1, stream_id=4)
    elif family == "ce_adamw":
        dec, wv = bf(16384, 512), bf(1536, 512)
        tgt = torch.randint(0, 1536, (16384,), device=dev)
        p = torch.randn(1 << 24, device=dev); g = torch.randn_like(p); m = torch.zeros_like(p); vv = torch.zeros_like(p)
        lr, st = torch.full((1,), 1e-3, device=dev), torch.zeros(1, device=dev, dtype=torch.int32)
        ids = torch.randint(0, 1536, (M,), device=dev); dx = torch.randn(M, 512, device=dev); tab = torch.zeros(1536, 512, device=dev)
        def fn():
            lib.lmhead_cross_entropy(dec, wv, tgt)
            lib.adamw_step(p, g, m, vv, lr, st)
            lib.embed_bwd(ids, dx, tab, 1024, shift=True)
    else:
        a = torch.randn(4096, 4096, device=dev).bfloat16()
        fn = lambda: a @ a
    if build_only:
        return fn
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    print("aggressor %s: %d launches in %.0f s" % (family, n, time.time() - t0), flush=True)


def inprocess(family, seconds):
    """victim and aggressor in ONE process on two streams; returns (victim launches, launches with a wrong frame,
    aggressor launches, fraction of the victim's time an aggressor kernel was queued beside it)."""
    sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
    import torch
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio
    dev = torch.device("cuda:0")
    fn = aggressor(family, 0, build_only=True)
    audio = torch.from_numpy(synth_audio(2, seed=51)).to(dev)
    ref = sp.logmel_segments(audio, out_bf16=True).clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    nbad = torch.zeros((), device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    n = na = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        # keep stream B full for the whole burst of victim launches: ~40 aggressor kernels of 50-400 us beside 400 victim
        # launches (+ their compare kernels) of ~15 us
        with torch.cuda.stream(sb):
            for _ in range(40):
                fn()
        with torch.cuda.stream(sa):
            for _ in range(400):
                out = sp.logmel_segments(audio, out_bf16=True)
                nbad += (out != ref).any()
        n += 400
        na += 40
        sa.synchronize()
        sb.synchronize()
    return n, int(nbad.item()), na



def victim(seconds, shift_gb=0):
    sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
    import torch
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio
    dev = torch.device("cuda:0")
    # shift_gb > 0: allocate that much FIRST, so that every later buffer of this process sits at other virtual addresses
    # than the aggressor's first buffers (two torch processes otherwise hand out identical addresses)
    pad = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(int(shift_gb))]
    audio = torch.from_numpy(synth_audio(2, seed=51)).to(dev)
    ref = sp.logmel_segments(audio, out_bf16=True).clone()
    bad = n = 0
    nbad = torch.zeros((), device=dev, dtype=torch.int64)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(200):
            out = sp.logmel_segments(audio, out_bf16=True)
            nbad += (out != ref).any()
        n += 200
    print("victim (first buffer at %#x): %d log-mel launches, %d with a wrong frame" % (audio.data_ptr(), n, int(nbad.item())), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--inprocess":
        secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
        fams = sys.argv[3].split(",") if len(sys.argv) > 3 else FAMILIES
        tot = 0
        for fam in fams:
            n, bad, na = inprocess(fam, secs)
            tot += n
            print("== one process, stream B = %-18s victim (stream A): %d log-mel launches, %d with a wrong frame; %d aggressor launches"
                  % (fam, n, bad, na), flush=True)
        print("total victim launches: %d" % tot)
    elif sys.argv[1] == "--aggressor":
        aggressor(sys.argv[2], float(sys.argv[3]))
    elif sys.argv[1] == "--victim":
        victim(float(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
        fams = sys.argv[2].split(",") if len(sys.argv) > 2 else FAMILIES
        shift = sys.argv[3] if len(sys.argv) > 3 else "0"
        me = os.path.abspath(__file__)
        for fam in fams:
            a = subprocess.Popen([sys.executable, me, "--aggressor", fam, str(secs + 25)])
            time.sleep(22)                          # the aggressor is past its imports and allocations
            v = subprocess.run([sys.executable, me, "--victim", str(secs), shift], capture_output=True, text=True)
            print("== beside %-18s %s" % (fam, (v.stdout.strip().splitlines() or ["(no output) " + v.stderr[-300:]])[-1]), flush=True)
            a.wait()
