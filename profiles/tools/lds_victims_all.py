"""Which kernels are VICTIMS of the co-residency fault?  profiles/tools/lds_victim.py showed the round-1 log-mel
kernel computing wrong frames whenever flash-attention forward (or another LDS-DMA + MFMA kernel that leaves LDS free on its
CU) runs beside it on another stream.  Here every kernel family of the training step takes the victim's place: it loops on
stream A, each launch's outputs compared bit for bit with its first launch, while attn_fwd loops on stream B of the same
process.     python3 profiles/tools/lds_victims_all.py [seconds per family = 8] [aggressor = attn_fwd]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mr-mt3_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from mrmt3 import lib
import lds_victim

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
agg_name = sys.argv[2] if len(sys.argv) > 2 else "attn_fwd"
dev = torch.device("cuda:0")
lib.load()
M = 8192
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()


def victims():
    v = {}
    x, y, w = torch.randn(M, 512, device=dev), bf(M, 512), torch.ones(512, device=dev)
    v["add_rmsnorm_fwd"] = lambda: lib.add_rmsnorm_fwd(x, y, w, 1e-6, torch.bfloat16, p=0.1, seed=1, stream_y=3)[1:]
    x1 = torch.randn(M, 512, device=dev)
    rstd = torch.rsqrt((x1 * x1).mean(-1) + 1e-6)
    dres = bf(M, 512)

    def nb():
        dw = torch.zeros(512, device=dev)
        dx1, dy = lib.add_rmsnorm_bwd(y, dres, x1, rstd, w, dw, p=0.1, seed=1, stream_y=3, dx1_dtype=torch.bfloat16)
        return dx1, dy, dw
    v["add_rmsnorm_bwd(+dw)"] = nb
    h, dg = bf(M, 2048), bf(M, 1024)
    v["geglu_bwd"] = lambda: (lib.geglu_bwd(h, dg, p=0.1, seed=1, stream_id=4),)
    dec, wv = bf(M, 512), bf(1536, 512)
    tgt = torch.randint(0, 1536, (M,), device=dev)
    v["lmhead_ce"] = lambda: lib.lmhead_cross_entropy(dec, wv, tgt)
    ids = torch.randint(0, 1536, (M,), device=dev)
    dx = torch.randn(M, 512, device=dev)

    def eb():
        tab = torch.zeros(1536, 512, device=dev)
        lib.embed_bwd(ids, dx, tab, 1024, shift=True)
        return (tab,)
    v["embed_bwd"] = eb
    a, b = bf(M, 512), bf(1152, 512)
    os.environ["MRMT3_GEMM8"] = "1"
    v["gemm_nt8"] = lambda: (lib.gemm_nt(a, b),)
    a2, b2 = bf(1024, 512), bf(384, 512)
    v["gemm_nt_tile"] = lambda: (lib.gemm_nt(a2, b2),)
    v["gemm_nt_geglu"] = lambda: lib.gemm_nt_geglu(a, bf(2048, 512) if False else wgl, p=0.1, seed=1, stream_id=2)
    ga, gb = bf(M, 512), bf(M, 384)

    def tn():
        o = torch.zeros(512, 384, device=dev)
        lib.gemm_tn(ga, gb, o)
        return (o,)
    v["gemm_tn"] = tn
    wr = bf(512, 384)
    ar = bf(M, 384)
    v["gemm_nt_addnorm"] = lambda: lib.gemm_nt_addnorm(ar, wr, x, w, 1e-6, p=0.1, seed=1, stream_y=3)
    from contrib import spectrograms as sp
    from mrmt3.synthetic import synth_audio
    audio = torch.from_numpy(synth_audio(2, seed=51)).to(dev)

    def old_logmel():
        os.environ["MRMT3_LOGMEL"] = "0"
        try:
            return (sp.logmel_segments(audio, out_bf16=True),)
        finally:
            os.environ.pop("MRMT3_LOGMEL", None)
    v["logmel (round-1 kernel)"] = old_logmel
    v["logmel (wave kernel)"] = lambda: (sp.logmel_segments(audio, out_bf16=True),)
    wtq, aq = bf(512, 1152), bf(M, 1152)

    def nbf():
        dw = torch.zeros(512, device=dev)
        dx1, dy = lib.gemm_nt_normbwd(aq, wtq, dres, x1, rstd, w, dw, p=0.1, seed=1, stream_y=3, dx1_dtype=torch.bfloat16)
        return dx1, dy, dw
    v["gemm_nt_normbwd"] = nbf
    v["gemm_nt_addnorm(p=0)"] = lambda: lib.gemm_nt_addnorm(ar, wr, x, w, 1e-6, p=0.0)
    for KK in (128, 256, 512, 1152):
        aK, wK = bf(M, KK), bf(512, KK)
        v["gemm_nt_addnorm(K=%d)" % KK] = (lambda aK=aK, wK=wK: lib.gemm_nt_addnorm(aK, wK, x, w, 1e-6, p=0.1, seed=1, stream_y=3))
    a384, wt384 = bf(M, 384), bf(512, 384)

    def nbf384():
        dw = torch.zeros(512, device=dev)
        return lib.gemm_nt_normbwd(a384, wt384, dres, x1, rstd, w, dw, p=0.1, seed=1, stream_y=3, dx1_dtype=torch.bfloat16) + (dw,)
    v["gemm_nt_normbwd(K=384)"] = nbf384
    v["gemm_nt(K=384,N=512)"] = lambda: (lib.gemm_nt(a384, wt384),)
    wt = bf(1024, 512)
    v["gemm_nt_geglubwd"] = lambda: (lib.gemm_nt_geglubwd(a, wt, h, p=0.1, seed=1, stream_id=4),)
    B, H, L = 4, 6, 1024
    q, k, vv = bf(B * L, 384), bf(B * L, 384), bf(B * L, 384)
    o, lse = lib.attn_fwd(q, k, vv, B, H, L, L, True, p=0.1, seed=1, stream_id=1)
    d_o = bf(B * L, 384)
    v["attn_fwd"] = lambda: lib.attn_fwd(q, k, vv, B, H, L, L, True, p=0.1, seed=1, stream_id=1)

    def ab():
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(vv)
        lib.attn_bwd(q, k, vv, o, d_o, lse, dq, dk, dv, B, H, L, L, True, p=0.1, seed=1, stream_id=1)
        return dq, dk, dv
    v["attn_bwd"] = ab
    kv = bf(B * 256, 768)
    o2, lse2, olo = lib.attn_fwd(q, kv[:, :384], kv[:, 384:], B, H, L, 256, False, p=0.1, seed=1, stream_id=1, want_lo=True)

    def ab1():
        os.environ["MRMT3_ATTN_ONEPASS_MIN_BH"] = "1"
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        lib.attn_bwd(q, kv[:, :384], kv[:, 384:], o2, d_o, lse2, dq, dkv[:, :384], dkv[:, 384:], B, H, L, 256, False, p=0.1, seed=1, stream_id=1, o_lo=olo)
        return dq, dkv
    v["attn_bwd_onepass"] = ab1
    p_ = torch.randn(1 << 22, device=dev)
    return v


wgl = bf(2048, 512)
agg = lds_victim.aggressor(agg_name, 0, build_only=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print("victim on stream A (every launch compared with its first), %s on stream B; %g s per family" % (agg_name, secs))
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for name, fn in victims().items():
    if only and name not in only:
        continue
    ref = [t.clone() for t in fn() if t is not None]
    torch.cuda.synchronize()
    nbad = torch.zeros((), device=dev, dtype=torch.int64)
    per = torch.zeros(len(ref), device=dev, dtype=torch.int64)
    n = 0
    t0 = time.time()
    while time.time() - t0 < secs:
        with torch.cuda.stream(sb):
            for _ in range(30):
                agg()
        with torch.cuda.stream(sa):
            for _ in range(100):
                out = [t for t in fn() if t is not None]
                bad = torch.zeros((), device=dev, dtype=torch.bool)
                for j, (t, r) in enumerate(zip(out, ref)):
                    a_, b_ = (t.view(torch.int16), r.view(torch.int16)) if t.dtype == torch.bfloat16 else (t, r)
                    bj = (a_ != b_).any()
                    per[j] += bj
                    bad = bad | bj
                nbad += bad
        n += 100
        sa.synchronize(); sb.synchronize()
    print("  victim %-22s %7d launches, %6d differ from the first  (per output tensor: %s)" % (name, n, int(nbad.item()), per.tolist()), flush=True)
