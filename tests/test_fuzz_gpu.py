"""Seeded random-shape sweeps over the kernels of the training step (the shape lists of test_kernels_gpu.py are hand-picked;
these walk the dispatch boundaries — 2047 / 2048 / 2049 rows, partial waves of tiles, N off the tile width, ragged
sequence lengths — with shapes nobody chose):

  * the NT product against an f32 torch product on sampled rows, with guard rows behind the output;
  * the weight-gradient product (single launch and the grouped launch of several ragged gradients) against f32 torch;
  * the MFMA attention kernels (forward, two-pass and one-pass backward) against the GENERAL attention kernels
    (attention_general.hip: exact-f32 arithmetic, one workgroup per row) WITH DROPOUT ON — both draw the same mask from
    (seed, stream, b, h, q, k), so this is a full-numerics comparison of two independent implementations under dropout,
    which torch cannot give (its generator is a different one).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mrmt3 import lib
    lib.load()
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


EDGE_M = [1, 7, 127, 128, 129, 255, 1023, 2047, 2048, 2049, 3072, 4095, 4096, 4097, 6144, 12288, 16383, 20001, 32768, 40960]


def _nt_shapes(n, seed):
    r = np.random.RandomState(seed)
    out = []
    for i in range(n):
        M = int(r.choice(EDGE_M)) if i % 3 else int(r.randint(1, 50000))
        N = int(r.choice([8, 64, 128, 136, 256, 384, 392, 512, 768, 1024, 1152, 1536, 2048, 2304]))
        K = 32 * int(r.randint(1, 49))
        od = ["bf16", "f32", "f32+"][int(r.randint(0, 3))]
        out.append((M, N, K, od))
    return out


@pytest.mark.parametrize("M,N,K,od", _nt_shapes(36, 11))
def test_gemm_nt_random_shapes_across_the_dispatch_boundaries(dev, M, N, K, od):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev).bfloat16()
    odt = torch.bfloat16 if od == "bf16" else torch.float32
    buf = torch.full((M + 3, N), 3.0, device=dev, dtype=odt)        # three guard rows behind the output
    out = buf[:M]
    lib.gemm_nt(a, b, out=out, accumulate=(od == "f32+"))
    rows = torch.randint(0, M, (min(M, 384),), generator=g).to(dev)
    rows = torch.cat([rows, torch.tensor([0, M - 1], device=dev)])
    ref = a[rows].float() @ b.float().t() + (3.0 if od == "f32+" else 0.0)
    err = (out[rows].float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
    assert err < (1e-2 if od == "bf16" else 2e-6), (M, N, K, od, err)
    assert (buf[M:] == 3.0).all(), "rows behind the output were written"
    out2 = torch.empty(M, N, device=dev, dtype=odt)
    if od != "f32+":
        lib.gemm_nt(a, b, out=out2)
        assert torch.equal(out2, out)                                   # same bits in a different buffer


def _tn_shapes(n, seed):
    r = np.random.RandomState(seed)
    out = [(int(r.choice(EDGE_M)) if i % 2 else int(r.randint(1, 70000)), 8 * int(r.randint(1, 260)), 8 * int(r.randint(1, 130)))
           for i in range(n)]
    # widths on the 128 / 64 grid but off the 256-wide tile (the ping-pong kernel's shifted last tile) and just off that grid
    return out + [(32768, 640, 576), (20001, 1152, 832), (8200, 896, 320), (33000, 2048, 960), (32768, 576, 512),
                  (32768, 512, 520), (34773, 696, 1008), (32768, 2040, 976)]


@pytest.mark.parametrize("M,N1,N2", _tn_shapes(16, 5))     # (the last two are the shapes that found the tn8 admission bug)
def test_gemm_tn_random_shapes(dev, M, N1, N2):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(M + N1 * 3 + N2)
    a = torch.randn(M, N1, generator=g).to(dev).bfloat16()
    b = torch.randn(M, N2, generator=g).to(dev).bfloat16()
    ref = a.float().t() @ b.float()
    out = torch.full((N1, N2), 7.0, device=dev)
    lib.gemm_tn(a, b, out)
    assert _rel(out, ref) < 1e-5, (M, N1, N2)
    out2 = torch.empty(N1, N2, device=dev)
    lib.gemm_tn(a, b, out2)
    assert torch.equal(out, out2)


def test_grouped_weight_gradients_of_random_ragged_shapes(dev):
    """One grouped launch (mrmt3_tn_group_plan / _run) over twelve gradients whose token counts and widths are random:
    every one equals the f32 product; a second plan + run gives the same bits."""
    from mrmt3 import lib
    r = np.random.RandomState(3)
    items = []
    for i in range(12):
        M = int(r.choice([1024, 1536, 3072, 4096, 12288, 16384])) if i % 2 else 1024 + 8 * int(r.randint(0, 2000))
        N1, N2 = 128 * int(r.randint(1, 10)), 128 * int(r.randint(1, 6))
        a = torch.randn(M, N1, device=dev).bfloat16()
        b = torch.randn(M, N2, device=dev).bfloat16()
        items.append((a, b, torch.zeros(N1, N2, device=dev)))
    outs = []
    n_grouped = 0
    for _ in range(2):
        grp = lib.TnGroup()
        for a, b, o in items:
            if lib.TnGroup.ok(a, b, o):
                grp.add(a, b, o, accumulate=False)
                n_grouped += 1
            else:
                lib.gemm_tn(a, b, o)
        grp.flush()
        torch.cuda.synchronize()
        outs.append([o.clone() for _, _, o in items])
    assert n_grouped >= 12
    for (a, b, _), o, o2 in zip(items, outs[0], outs[1]):
        assert _rel(o, a.float().t() @ b.float()) < 1e-5
        assert torch.equal(o, o2)


def _attn_cases(n, seed):
    r = np.random.RandomState(seed)
    out = [(16, 6, 300, 256, False, 0.1), (24, 4, 256, 256, False, 0.1), (4, 6, 1024, 1024, True, 0.1)]   # one-pass sites + the causal square
    while len(out) < n:
        causal = bool(r.randint(0, 2))
        Lq = int(r.randint(1, 700))
        Lk = Lq if causal else int(r.randint(1, 700))
        out.append((int(r.randint(1, 5)), int(r.randint(1, 7)), Lq, Lk, causal, float(r.choice([0.0, 0.1, 0.25]))))
    return out


@pytest.mark.parametrize("B,H,Lq,Lk,causal,p", _attn_cases(14, 9))
def test_mfma_attention_agrees_with_the_general_kernels_under_dropout(dev, B, H, Lq, Lk, causal, p):
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Lq * 7 + Lk)
    q = (torch.randn(B * Lq, H * 64, generator=g) * 0.35).to(dev).bfloat16()
    k = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    v = torch.randn(B * Lk, H * 64, generator=g).to(dev).bfloat16()
    d_o = torch.randn(B * Lq, H * 64, generator=g).to(dev).bfloat16()
    seed, stream = 1234 + Lq, 5
    lib.dispatch_counts(reset=True)
    o, lse, o_lo = lib.attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream, want_lo=True)
    # the general kernels on f32 copies of the same bf16 values: an exact-arithmetic reference with the same mask
    qf, kf, vf, df = q.float(), k.float(), v.float(), d_o.float()
    of, lf = lib.attn_fwd_bias(qf, kf, vf, None, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream)
    assert torch.allclose(lse, lf, atol=2e-3), (lse - lf).abs().max().item()
    assert _rel(o, of) < 6e-3, _rel(o, of)
    # a dropped element is exactly absent from both: with V = const the row sums expose the kept SET through O; here the
    # full-numerics agreement at 6e-3 is already far below one wrong mask bit's effect when rows are short
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream, o_lo=o_lo)
    rq, rk, rv, _ = lib.attn_bwd_bias(qf, kf, vf, of, df, lf, None, B, H, Lq, Lk, causal, p=p, seed=seed, stream_id=stream)
    c = lib.dispatch_counts()
    assert c["attn_fwd"] == 1 and c["attn_bwd"] + c["attn_bwd_onepass"] == 1 and c["attn_f32"] == 2, c
    for got, ref, name in ((dq, rq, "dq"), (dk, rk, "dk"), (dv, rv, "dv")):
        tol = 1.2e-2 if name != "dv" else 8e-3
        assert _rel(got, ref) < tol or (got.float() - ref).abs().max() < 1e-4, (name, _rel(got, ref), c)
    if Lk == 256 and not causal and B * H >= 96:
        assert c["attn_bwd_onepass"] == 1, c
