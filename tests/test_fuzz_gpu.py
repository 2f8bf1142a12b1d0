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


@pytest.mark.parametrize("M,N,K,od", [(3072, 512, 2048, "bf16"), (3072, 512, 6144, "f32+"), (3000, 512, 2048, "bf16"),
                                      (4096, 1024, 2048, "f32"), (1024, 256, 4096, "bf16"), (2049, 384, 3072, "bf16"),
                                      (6144, 512, 2048, "f32"), (3072, 512, 2304, "bf16")])
def test_gemm_nt_split_over_k_for_short_inputs(dev, M, N, K, od):
    """mrmt3_gemm_nt_ws: short inputs with a long K (the encoder's rows at 12 segments per GPU) run as 2-4 K ranges of one
    launch + a reduce in split order.  Against f32 torch, bitwise repeatable, guard rows untouched, the dispatch
    counter says the split form ran, and the plain entry point (no workspace) gives the same values to bf16 rounding."""
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev).bfloat16()
    odt = torch.bfloat16 if od == "bf16" else torch.float32
    assert lib.load().mrmt3_gemm_nt_workspace_bytes(M, N, K, 1) > 0, "not a split-K shape"
    buf = torch.full((M + 3, N), 3.0, device=dev, dtype=odt)
    out = buf[:M]
    lib.dispatch_counts(reset=True)
    lib.gemm_nt(a, b, out=out, accumulate=(od == "f32+"))
    assert lib.dispatch_counts()["gemm_nt_splitk"] == 1
    ref = a.float() @ b.float().t() + (3.0 if od == "f32+" else 0.0)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (1e-2 if od == "bf16" else 5e-6), err
    assert (buf[M:] == 3.0).all()
    out2 = torch.full((M, N), 3.0, device=dev, dtype=odt)
    lib.gemm_nt(a, b, out=out2, accumulate=(od == "f32+"))
    assert torch.equal(out, out2)
    # the plain entry point on the same operands
    out3 = torch.full((M, N), 3.0, device=dev, dtype=odt)
    L = lib.load()
    rc = L.mrmt3_gemm_nt(lib._p(a), a.stride(0), lib._p(b), b.stride(0), lib._p(out3), out3.stride(0), M, N, K, 1,
                         1 if od == "bf16" else 0, int(od == "f32+"), lib._stream())
    assert rc == 0
    assert (out3.float() - out.float()).abs().max().item() / ref.abs().max().item() < (1e-2 if od == "bf16" else 5e-6)


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


# ---- row-wise kernels at row counts nobody chose ------------------------------------------------------------------------
ROWS = [1, 3, 31, 33, 257, 1000, 2047, 2049, 3072, 12289, 16384, 40001]


def _gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("cols,ydt,rdt", [(512, torch.bfloat16, torch.bfloat16), (512, torch.float32, torch.float32),
                                          (256, torch.bfloat16, torch.float32), (1024, torch.float32, torch.float32),
                                          (2048, torch.bfloat16, torch.float32)])
def test_add_rmsnorm_random_rows(dev, rows, cols, ydt, rdt):
    """x1 = x0 + y, xn = RMSNorm(x1) w and its backward (incl. the norm-weight gradient's partial rows) at every row count,
    both residual-gradient stream types, against autograd in f32."""
    from mrmt3 import lib
    if rows > 20000 and cols > 512:
        pytest.skip("large case only at the model width")
    g = torch.Generator(device="cpu").manual_seed(rows * 13 + cols)
    x0 = torch.randn(rows, cols, generator=g).to(dev)
    y = torch.randn(rows, cols, generator=g).to(dev).to(ydt)
    w = (1 + 0.1 * torch.randn(cols, generator=g)).to(dev)
    x1, xn, rstd = lib.add_rmsnorm_fwd(x0, y, w, 1e-6, torch.bfloat16)
    x1r = (x0 + y.float()).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    xnr = wr * (x1r * torch.rsqrt(x1r.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert torch.allclose(x1, x1r, atol=1e-6) and _rel(xn, xnr) < 4e-3
    assert torch.allclose(rstd, torch.rsqrt(x1r.detach().pow(2).mean(-1) + 1e-6), rtol=1e-5, atol=1e-7)
    dxn = torch.randn(rows, cols, generator=g).to(dev).bfloat16()
    dres = torch.randn(rows, cols, generator=g).to(dev).to(rdt)
    (xnr * dxn.float()).sum().backward()
    dw = torch.zeros(cols, device=dev)
    dx1, dy = lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, w, dw, dx1_dtype=rdt)
    want = x1r.grad + dres.float()
    assert _rel(dx1, want) < (4e-3 if rdt == torch.bfloat16 else 1e-5)
    assert _rel(dy, want) < 4e-3
    assert _rel(dw, wr.grad) < 2e-5, _rel(dw, wr.grad)


@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("dff", [1024, 8, 200, 1536])
def test_geglu_random_rows_and_widths(dev, rows, dff):
    from mrmt3 import lib
    if rows > 20000 and dff != 1024:
        pytest.skip("large case only at the model width")
    g = torch.Generator(device="cpu").manual_seed(rows + dff)
    hb = torch.randn(rows, 2 * dff, generator=g).to(dev).bfloat16()
    hr = hb.float().requires_grad_(True)
    gr = _gelu_new(hr[:, :dff]) * hr[:, dff:]
    gb = lib.geglu_fwd(hb)
    assert _rel(gb, gr) < 4e-3
    dg = torch.randn(rows, dff, generator=g).to(dev).bfloat16()
    (gr * dg.float()).sum().backward()
    assert _rel(lib.geglu_bwd(hb, dg), hr.grad) < 4e-3
    # dropout: forward and backward draw the same mask, kept elements scaled by 1 / (1 - p)
    gd = lib.geglu_fwd(hb, p=0.1, seed=7, stream_id=3)
    kept = gd != 0
    frac = kept.float().mean().item()
    if rows * dff > 20000:
        assert abs(frac - 0.9) < 0.02, frac
    big = gb.float().abs() > 1e-2
    assert torch.allclose(gd.float()[kept & big], (gb.float() / 0.9)[kept & big], rtol=1.2e-2)
    dhd = lib.geglu_bwd(hb, dg, p=0.1, seed=7, stream_id=3)
    dropped_rows_cols = (~kept) & big
    assert (dhd[:, :dff][dropped_rows_cols] == 0).all() and (dhd[:, dff:][dropped_rows_cols] == 0).all()


@pytest.mark.parametrize("rows,dff,K", [(4096, 1024, 512), (4097, 1024, 512), (12288, 1024, 512), (2048, 512, 256),
                                        (20001, 1024, 512), (5000, 384, 128), (3000, 1024, 512), (1, 1024, 512)])
def test_fused_wi_geglu_random_shapes_equal_the_two_kernels(dev, rows, dff, K):
    """mrmt3_gemm_nt_geglu (projection + gated GELU + dropout in one launch, or its unfused fallback for shapes the
    ping-pong kernel does not take) against gemm_nt followed by geglu_fwd: the same bits, dropout on."""
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(rows + dff + K)
    x = torch.randn(rows, K, generator=g).to(dev).bfloat16()
    wi = (torch.randn(2 * dff, K, generator=g) * 0.05).to(dev).bfloat16()
    h, gg = lib.gemm_nt_geglu(x, wi, p=0.1, seed=11, stream_id=2)
    h2 = lib.gemm_nt(x, wi)
    g2 = lib.geglu_fwd(h2, p=0.1, seed=11, stream_id=2)
    assert torch.equal(h, h2) and torch.equal(gg, g2)
    ref = x.float() @ wi.float().t()
    assert _rel(h, ref) < 4e-3


@pytest.mark.parametrize("rows,V,chunk", [(1, 1536, 16384), (1000, 1536, 256), (2049, 1536, 2048), (16385, 1536, 16384),
                                          (40001, 1536, 16384), (777, 512, 100), (3000, 2048, 1024)])
@pytest.mark.parametrize("weighted", [False, True])
def test_lmhead_ce_random_rows_vocabularies_and_chunks(dev, rows, V, chunk, weighted):
    """lm_head + cross entropy over row chunks against torch: loss (mean over non-ignored rows, or the reference's
    instrument-weighted form), dlogits, ragged last chunk, ignore_index rows."""
    from mrmt3 import lib
    g = torch.Generator(device="cpu").manual_seed(rows + V)
    dec = torch.randn(rows, 512, generator=g).to(dev).bfloat16()
    w = (torch.randn(V, 512, generator=g) * 0.05).to(dev).bfloat16()
    tg = torch.randint(0, V, (rows,), generator=g).to(dev)
    tg[::5] = -100
    if rows == 1:
        tg[0] = 3
    inst_lo, inst_hi = V // 2, V // 2 + 100
    loss, dl = lib.lmhead_cross_entropy(dec, w, tg, grad_dtype=torch.float32, weighted=weighted, inst_lo=inst_lo,
                                        inst_hi=inst_hi, chunk_rows=chunk)
    logits = (dec.float() @ w.float().t()).double().requires_grad_(True)
    valid = tg != -100
    per_row = torch.nn.functional.cross_entropy(logits, tg.clamp(min=0), reduction="none")
    if weighted:
        # the reference's weighted loss (tasks/mt3_net.py:75-165) as restated — and pinned — in oracle/t5_ref.py
        from oracle import t5_ref
        ref = t5_ref.weighted_ce_loss(logits[None], tg[None], lo=inst_lo, hi=inst_hi)
    else:
        ref = (per_row * valid).sum() / valid.sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    assert torch.allclose(dl.double(), logits.grad, atol=1e-7, rtol=2e-4)


# ---- the whole model at shapes nobody chose -------------------------------------------------------------------------------
def _build(variant, dtype, dev):
    from mrmt3.synthetic import T5_SMALL
    if variant == "t5":
        from models.t5 import T5ForConditionalGeneration as M
        m = M(T5_SMALL, compute_dtype=dtype)
    else:
        import importlib
        mod, cls = {"segmem_v1": ("models.t5_segmem", "T5SegMem"), "segmem_v2": ("models.t5_segmem_v2", "T5SegMemV2"),
                    "segmem_v2_with_prev": ("models.t5_segmem_v2_with_prev", "T5SegMemV2WithPrev")}[variant]
        m = getattr(importlib.import_module(mod), cls)(T5_SMALL, segmem_num_layers=1, segmem_length=64,
                                                       compute_dtype=dtype)
    return m.load_golden().to(dev).eval()


@pytest.mark.parametrize("variant,B,frames,Ld", [("t5", 3, 256, 72), ("segmem_v2_with_prev", 1, 200, 200),
                                                 ("segmem_v1", 5, 256, 104), ("segmem_v2", 3, 136, 136),
                                                 ("t5", 7, 64, 8), ("segmem_v2_with_prev", 4, 256, 328)])
def test_model_gradients_at_odd_batch_sizes_and_lengths_match_the_oracle(dev, variant, B, frames, Ld):
    """Odd batch sizes, encoder lengths off the 256-frame grid and target lengths off every tile size: the fp32 engine
    (exact-f32 kernels) against oracle/t5_ref.py autograd to rel-L2 1e-4 per gradient tensor, and the bf16 engine
    (MFMA kernels) to bf16 noise (cosine > 0.999, rel-L2 < 6e-2; loss within 1e-3 + 2e-2 / sqrt(valid tokens))."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel, synth_labels
    from oracle import t5_ref
    torch.set_num_threads(8)
    mel = torch.from_numpy(synth_mel(B, frames=frames, seed=B + frames))
    lab = torch.from_numpy(synth_labels(B, Ld, full=False, seed=70 + Ld, mean_len=max(2, Ld // 2)))
    prev = torch.from_numpy(synth_labels(B, Ld, full=False, seed=90 + Ld, mean_len=max(2, Ld // 2)))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in golden_weights(T5_SMALL, 0 if variant == "t5" else 1).items()}
    logits = t5_ref.forward_logits(sd, T5_SMALL, mel, lab, variant=variant, targets_prev=prev.clone())
    ref_loss = t5_ref.ce_loss(logits, lab)
    ref_loss.backward()
    for dt in (torch.float32, torch.bfloat16):
        m = _build(variant, dt, dev)
        out = m(inputs=mel.to(dev), labels=lab.to(dev), targets_prev=prev.clone().to(dev))
        assert out.shape == logits.shape
        loss = torch.nn.functional.cross_entropy(out.float().view(-1, 1536), lab.to(dev).view(-1), ignore_index=-100)
        m.flat.ensure_grads()
        m.flat.G.zero_()
        loss.backward()
        # bf16: north_star's 1e-3 is stated for a full batch (65 536 target tokens); the mean over the n valid tokens of
        # these small cases carries the per-token logit noise (<= 3e-2, test_model_gpu.py) divided by sqrt(n)
        n_valid = int((lab != -100).sum())
        tol = 2e-5 if dt == torch.float32 else 1e-3 + 2e-2 / n_valid ** 0.5
        assert abs(loss.item() - ref_loss.item()) < tol, (dt, loss.item(), ref_loss.item(), n_valid)
        worst = (0.0, "")
        for k, ref in sd.items():
            g = m.flat.grad(k).float().cpu()
            r = ref.grad
            if r is None or r.norm() == 0:
                assert g.norm() < 1e-6, k
                continue
            rel = ((g - r).norm() / r.norm()).item()
            worst = max(worst, (rel, k))
            if dt == torch.float32:
                assert rel < 1e-4, (k, rel)
            else:
                cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
                assert cos > 0.999 and rel < 6e-2, (k, cos, rel)
        print(variant, B, frames, Ld, dt, "worst rel-L2 %.3e (%s)" % worst)
        del m


@pytest.mark.parametrize("variant,B,frames,ml", [("t5", 5, 200, 37), ("t5", 11, 256, 21), ("segmem_v2_with_prev", 3, 136, 80),
                                                 ("segmem_v2", 7, 72, 70), ("t5", 1, 8, 70)])
def test_greedy_decode_at_odd_batches_and_encoder_lengths_is_the_oracles_token_ids(dev, variant, B, frames, ml):
    """fp32 greedy decode (KV cache, hipGraph replay, early EOS) at batch sizes and encoder lengths off every tile against
    the oracle's cache-free loop (models/t5.py:251-302 restated): the same token ids."""
    from mrmt3.synthetic import T5_SMALL, golden_weights, synth_mel
    from oracle import t5_ref
    torch.set_num_threads(8)
    w = golden_weights(T5_SMALL, 0 if variant == "t5" else 1)
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][1] *= 2.6          # EOS competitive: some rows finish early, others run to the end
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    mel = torch.from_numpy(synth_mel(B, frames=frames, seed=B * 7 + frames))
    gen = {"t5": t5_ref.generate_t5,
           "segmem_v2": lambda *a, **k: t5_ref.generate_segmem_v2(*a, with_prev=False, **k),
           "segmem_v2_with_prev": lambda *a, **k: t5_ref.generate_segmem_v2(*a, with_prev=True, **k)}[variant]
    with torch.no_grad():
        ref = gen(sd, T5_SMALL, mel, max_length=ml)
    m = _build(variant, torch.float32, dev)
    with torch.no_grad():
        m.flat.load_numpy(w)
    ids = m.generate(mel.to(dev), max_length=ml)
    assert torch.equal(ids.cpu(), ref), (ids.cpu(), ref)
