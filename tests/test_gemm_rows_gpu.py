"""The fused "projection + row kernel" launches of csrc/gemm_rows.hip against the two kernels they replace, through the C
ABI: mrmt3_gemm_nt_addnorm (o / co / wo projection -> residual add + dropout + T5LayerNorm, HF T5LayerSelfAttention /
T5LayerFF as called at models/t5.py:636-648), mrmt3_gemm_nt_normbwd (data gradient -> backward of that norm),
mrmt3_gemm_nt_geglubwd (wo data gradient -> gated-GELU backward).  The forward fusion and the GEGLU backward promise the
SAME BITS as the two-kernel form (the tile is rounded to bf16 exactly as the stand-alone product writes it and the row
arithmetic is restated in the same order); the norm backward promises the same bits for dx1 / dy and the same norm-weight
gradient up to the grouping of its f32 partial sums (64 rows per partial row instead of 32).  Each case is also held to an
f32 torch reference of the whole chain, so that the pair cannot be wrong together.

Every case runs under BOTH tile heights (the knob MRMT3_ROWS_BM = 64 / 128): the launch picks 128-row tiles by itself only
from 32 641 rows on (one workgroup per CU), which is what the 64-segment benchmark step dispatches for the decoder's
65 536 rows — the `*_at_the_benchmark_row_count` cases take that path WITHOUT the knob and assert that they did."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mr-mt3_amd"))
    from mrmt3 import lib as L
    L.load()
    return L


@pytest.fixture(params=[64, 128], ids=["tile64", "tile128"])
def tile(request, knobs):
    """Rows of the fused kernels' tile, forced (gemm_rows.hip: gr_bm): GRCfg<64> is 2 activation slots / one piece per wave /
    8-row row phases, GRCfg<128> 4 slots / 2 pieces / 16-row row phases with other counted waits — separate instantiations."""
    knobs.set("MRMT3_ROWS_BM", request.param)
    return request.param


def _rand(shape, scale, seed, dtype=torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(dtype)


# (rows, K): the o / co projection (K = 384), wo (K = 1024), ragged rows (not a multiple of 64), one partial tile
FWD_SHAPES = [(4096, 384), (4096, 1024), (1000, 384), (40, 128), (16384, 512)]


@pytest.mark.parametrize("rows,K", FWD_SHAPES)
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("out_drop", [False, True])
def test_gemm_nt_addnorm_equals_the_two_kernels_bitwise(lib, tile, rows, K, p, out_drop):
    a = _rand((rows, K), 1.0, 1)
    w = _rand((512, K), K ** -0.5, 2)
    x0 = _rand((rows, 512), 1.0, 3, torch.float32)
    wn = (1.0 + 0.1 * torch.randn(512, device="cuda")).float()
    step = torch.tensor([7], device="cuda", dtype=torch.int32)
    kw = dict(p=p, seed=365, stream_y=11, stream_out=12, out_drop=out_drop, step=step)
    y = lib.gemm_nt(a, w, out_dtype=torch.bfloat16)
    x1_ref, xn_ref, rstd_ref = lib.add_rmsnorm_fwd(x0, y, wn, 1e-6, torch.bfloat16, **kw)
    before = lib.dispatch_counts()["gemm_nt_addnorm"]
    x1, xn, rstd = lib.gemm_nt_addnorm(a, w, x0, wn, 1e-6, **kw)
    assert lib.dispatch_counts()["gemm_nt_addnorm"] == before + 1
    torch.cuda.synchronize()
    assert torch.equal(x1, x1_ref), (x1 - x1_ref).abs().max().item()
    assert torch.equal(rstd, rstd_ref), (rstd - rstd_ref).abs().max().item()
    assert torch.equal(xn.view(torch.int16), xn_ref.view(torch.int16)), (xn.float() - xn_ref.float()).abs().max().item()
    if p == 0.0:
        # f32 reference of the chain (the product rounded to bf16 like y_dtype does)
        yf = (a.float() @ w.float().t()).bfloat16().float()
        xf = x0 + yf
        ref = wn * xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6)
        # y is a bf16 rounding of differently ordered f32 sums: the odd element is one bf16 step of y off
        assert (x1 - xf).abs().max().item() <= 2.0 ** -7 * max(1.0, yf.abs().max().item())
        assert (xn.float() - ref).abs().max().item() <= 0.05 * ref.abs().max().item()
        assert ((xn.float() - ref).norm() / ref.norm()).item() < 4e-3


def test_gemm_nt_addnorm_in_place_residual_and_no_x1(lib, tile):
    rows, K = 2048, 384
    a, w = _rand((rows, K), 1.0, 4), _rand((512, K), K ** -0.5, 5)
    x0 = _rand((rows, 512), 1.0, 6, torch.float32)
    wn = torch.ones(512, device="cuda")
    ref = lib.gemm_nt_addnorm(a, w, x0, wn, 1e-6, p=0.1, seed=1, stream_y=3)
    xi = x0.clone()
    got = lib.gemm_nt_addnorm(a, w, xi, wn, 1e-6, p=0.1, seed=1, stream_y=3, x1=xi)           # x1 IS x0
    assert got[0].data_ptr() == xi.data_ptr() and torch.equal(xi, ref[0]) and torch.equal(got[1].view(torch.int16), ref[1].view(torch.int16))
    x0b = x0.clone()
    got = lib.gemm_nt_addnorm(a, w, x0b, wn, 1e-6, write_x1=False, p=0.1, seed=1, stream_y=3)   # the residual is not written
    assert torch.equal(x0b, x0) and torch.equal(got[1].view(torch.int16), ref[1].view(torch.int16))


BWD_SHAPES = [(4096, 1152), (4096, 384), (1000, 384), (16384, 2048), (72, 128)]   # (no shape whose stand-alone product runs split over K: another summation order)


@pytest.mark.parametrize("rows,K", BWD_SHAPES)
@pytest.mark.parametrize("res_in,res_out", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                            (torch.float32, torch.float32), (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gemm_nt_normbwd_equals_the_two_kernels(lib, tile, rows, K, res_in, res_out, p):
    a = _rand((rows, K), 1.0, 11)
    wt = _rand((512, K), K ** -0.5, 12)
    dres = _rand((rows, 512), 1.0, 13, res_in)
    x1 = _rand((rows, 512), 1.5, 14, torch.float32)
    rstd = torch.rsqrt((x1 * x1).mean(-1) + 1e-6)
    wn = (1.0 + 0.1 * torch.randn(512, device="cuda")).float()
    step = torch.tensor([3], device="cuda", dtype=torch.int32)
    kw = dict(p=p, seed=99, stream_y=21, step=step)
    dxn = lib.gemm_nt(a, wt, out_dtype=torch.bfloat16)
    dw_ref = torch.zeros(512, device="cuda")
    dx1_ref, dy_ref = lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, wn, dw_ref, dx1_dtype=res_out, **kw)
    dw = torch.zeros(512, device="cuda")
    before = lib.dispatch_counts()["gemm_nt_normbwd"]
    dx1, dy = lib.gemm_nt_normbwd(a, wt, dres, x1, rstd, wn, dw, dx1_dtype=res_out, **kw)
    assert lib.dispatch_counts()["gemm_nt_normbwd"] == before + 1
    assert lib.load().mrmt3_gemm_nt_normbwd_partial_rows(rows) == -(-rows // tile)
    torch.cuda.synchronize()
    assert dx1.dtype == res_out and torch.equal(dx1, dx1_ref), (dx1.float() - dx1_ref.float()).abs().max().item()
    assert torch.equal(dy.view(torch.int16), dy_ref.view(torch.int16))
    # the same sums grouped differently (64 rows per partial row instead of 32): f32 rounding of ~rows terms
    assert (dw - dw_ref).abs().max().item() <= 2e-5 * dw_ref.abs().max().item() + 1e-6, ((dw - dw_ref).abs().max().item(), dw_ref.abs().max().item())
    # in place on the residual gradient (the engine's form) and without dy
    if res_in == res_out:
        d2 = dres.clone()
        dx2, none = lib.gemm_nt_normbwd(a, wt, d2, x1, rstd, wn, None, want_dy=False, dx1=d2, **kw)
        assert none is None and dx2.data_ptr() == d2.data_ptr() and torch.equal(d2, dx1_ref)
    if p == 0.0 and res_out == torch.float32:
        g = (a.float() @ wt.float().t()).bfloat16().float()
        xh = x1 * rstd[:, None]
        gw = g * wn
        ref = rstd[:, None] * (gw - xh * (gw * xh).mean(-1, keepdim=True)) + dres.float()
        assert ((dx1 - ref).norm() / ref.norm()).item() < 3e-3
        dwf = (g * xh).sum(0)
        assert ((dw - dwf).norm() / dwf.norm()).item() < 3e-3


def test_gemm_nt_normbwd_partial_rows_feed_the_batched_reduce(lib, tile):
    """The engine's form: several sites leave their partial rows, ONE mrmt3_norm_dw_reduce sums them (a fused site next to
    a stand-alone site in the same batch)."""
    rows, K = 4096, 384
    batch = lib.NormDwBatch()
    a, wt = _rand((rows, K), 1.0, 31), _rand((512, K), K ** -0.5, 32)
    dres = _rand((rows, 512), 1.0, 33)
    x1 = _rand((rows, 512), 1.5, 34, torch.float32)
    rstd = torch.rsqrt((x1 * x1).mean(-1) + 1e-6)
    wn = torch.ones(512, device="cuda")
    dw_a, dw_b, dw_ref = (torch.zeros(512, device="cuda") for _ in range(3))
    dxn = lib.gemm_nt(a, wt, out_dtype=torch.bfloat16)
    lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, wn, dw_ref, dx1_dtype=torch.bfloat16)
    lib.gemm_nt_normbwd(a, wt, dres, x1, rstd, wn, dw_a, dx1_dtype=torch.bfloat16, defer=batch)
    lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, wn, dw_b, dx1_dtype=torch.bfloat16, defer=batch)
    assert dw_a.abs().max().item() == 0.0
    batch.flush()
    torch.cuda.synchronize()
    assert torch.equal(dw_b, dw_ref)
    assert (dw_a - dw_ref).abs().max().item() <= 2e-5 * dw_ref.abs().max().item()
    # twice in a row (the counters the reduce leaves behind are re-armed by the next launch)
    dw_c = torch.zeros(512, device="cuda")
    lib.gemm_nt_normbwd(a, wt, dres, x1, rstd, wn, dw_c, dx1_dtype=torch.bfloat16, defer=batch)
    batch.flush()
    assert torch.equal(dw_c, dw_a)


@pytest.mark.parametrize("rows", [4096, 1000, 24])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gemm_nt_geglubwd_equals_the_two_kernels_bitwise(lib, tile, rows, p):
    d, dff = 512, 1024
    dy = _rand((rows, d), 1.0, 41)
    wt = _rand((dff, d), d ** -0.5, 42)                     # wo^T
    h = _rand((rows, 2 * dff), 1.0, 43)
    step = torch.tensor([5], device="cuda", dtype=torch.int32)
    kw = dict(p=p, seed=7, stream_id=17, step=step)
    dg = lib.gemm_nt(dy, wt, out_dtype=torch.bfloat16)
    ref = lib.geglu_bwd(h, dg, **kw)
    before = lib.dispatch_counts()["gemm_nt_geglubwd"]
    got = lib.gemm_nt_geglubwd(dy, wt, h, **kw)
    assert lib.dispatch_counts()["gemm_nt_geglubwd"] == before + 1
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (got.float() - ref.float()).abs().max().item()
    if p == 0.0:
        hf = h.float().requires_grad_(True)
        g = torch.nn.functional.gelu(hf[:, :dff], approximate="tanh") * hf[:, dff:]
        g.backward((dy.float() @ wt.float().t()).bfloat16().float())
        assert ((got.float() - hf.grad).norm() / hf.grad.norm()).item() < 4e-3


# ---- the benchmark's own row count: 64 segments x 1024 tokens, tile height chosen by the launch itself ------------------

BENCH_ROWS = 65536


def _took_128_row_tiles(lib):
    """The launch's own choice at BENCH_ROWS rows (no knob): 128-row tiles once ceil(rows / 128) reaches the CU count."""
    return lib.load().mrmt3_gemm_nt_normbwd_partial_rows(BENCH_ROWS) == BENCH_ROWS // 128


@pytest.mark.parametrize("K", [384, 1024])          # o / co projection, wo
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gemm_nt_addnorm_at_the_benchmark_row_count(lib, knobs, K, p):
    knobs.unset("MRMT3_ROWS_BM")
    assert _took_128_row_tiles(lib), "a 256-CU MI355X runs 65 536 rows as 512 tiles of 128 rows"
    rows = BENCH_ROWS
    a, w = _rand((rows, K), 1.0, 51), _rand((512, K), K ** -0.5, 52)
    x0 = _rand((rows, 512), 1.0, 53, torch.float32)
    wn = (1.0 + 0.1 * torch.randn(512, device="cuda")).float()
    step = torch.tensor([9], device="cuda", dtype=torch.int32)
    kw = dict(p=p, seed=365, stream_y=5, stream_out=6, out_drop=True, step=step)
    y = lib.gemm_nt(a, w, out_dtype=torch.bfloat16)
    x1_ref, xn_ref, rstd_ref = lib.add_rmsnorm_fwd(x0, y, wn, 1e-6, torch.bfloat16, **kw)
    x1, xn, rstd = lib.gemm_nt_addnorm(a, w, x0, wn, 1e-6, **kw)
    torch.cuda.synchronize()
    assert torch.equal(x1, x1_ref) and torch.equal(rstd, rstd_ref) and torch.equal(xn.view(torch.int16), xn_ref.view(torch.int16))
    if p == 0.0:
        yf = (a.float() @ w.float().t()).bfloat16().float()
        xf = x0 + yf
        ref = wn * xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6)
        assert (x1 - xf).abs().max().item() <= 2.0 ** -7 * max(1.0, yf.abs().max().item())
        assert ((xn.float() - ref).norm() / ref.norm()).item() < 4e-3


@pytest.mark.parametrize("K", [384, 1152, 512])     # d_cq, d_qkv, and a K the engine sends to the ping-pong product today
@pytest.mark.parametrize("res", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gemm_nt_normbwd_at_the_benchmark_row_count(lib, knobs, K, res, p):
    knobs.unset("MRMT3_ROWS_BM")
    assert _took_128_row_tiles(lib)
    rows = BENCH_ROWS
    a, wt = _rand((rows, K), 1.0, 61), _rand((512, K), K ** -0.5, 62)
    dres = _rand((rows, 512), 1.0, 63, res)
    x1 = _rand((rows, 512), 1.5, 64, torch.float32)
    rstd = torch.rsqrt((x1 * x1).mean(-1) + 1e-6)
    wn = (1.0 + 0.1 * torch.randn(512, device="cuda")).float()
    step = torch.tensor([4], device="cuda", dtype=torch.int32)
    kw = dict(p=p, seed=99, stream_y=21, step=step)
    dxn = lib.gemm_nt(a, wt, out_dtype=torch.bfloat16)
    dw_ref, dw = torch.zeros(512, device="cuda"), torch.zeros(512, device="cuda")
    dx1_ref, dy_ref = lib.add_rmsnorm_bwd(dxn, dres, x1, rstd, wn, dw_ref, dx1_dtype=res, **kw)
    before = lib.dispatch_counts()["gemm_nt_normbwd"]
    dx1, dy = lib.gemm_nt_normbwd(a, wt, dres, x1, rstd, wn, dw, dx1_dtype=res, **kw)
    assert lib.dispatch_counts()["gemm_nt_normbwd"] == before + 1
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx1_ref) and torch.equal(dy.view(torch.int16), dy_ref.view(torch.int16))
    assert (dw - dw_ref).abs().max().item() <= 2e-5 * dw_ref.abs().max().item() + 1e-6
    if p == 0.0 and res == torch.float32:
        g = (a.float() @ wt.float().t()).bfloat16().float()
        xh = x1 * rstd[:, None]
        gw = g * wn
        ref = rstd[:, None] * (gw - xh * (gw * xh).mean(-1, keepdim=True)) + dres.float()
        assert ((dx1 - ref).norm() / ref.norm()).item() < 3e-3
        dwf = (g * xh).sum(0)
        assert ((dw - dwf).norm() / dwf.norm()).item() < 3e-3


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_gemm_nt_geglubwd_at_the_benchmark_row_count(lib, knobs, p):
    knobs.unset("MRMT3_ROWS_BM")
    assert _took_128_row_tiles(lib)
    rows, d, dff = BENCH_ROWS, 512, 1024
    dy, wt = _rand((rows, d), 1.0, 71), _rand((dff, d), d ** -0.5, 72)
    h = _rand((rows, 2 * dff), 1.0, 73)
    step = torch.tensor([5], device="cuda", dtype=torch.int32)
    kw = dict(p=p, seed=7, stream_id=17, step=step)
    ref = lib.geglu_bwd(h, lib.gemm_nt(dy, wt, out_dtype=torch.bfloat16), **kw)
    before = lib.dispatch_counts()["gemm_nt_geglubwd"]
    got = lib.gemm_nt_geglubwd(dy, wt, h, **kw)
    assert lib.dispatch_counts()["gemm_nt_geglubwd"] == before + 1
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    if p == 0.0:
        hf = h.float().requires_grad_(True)
        g = torch.nn.functional.gelu(hf[:, :dff], approximate="tanh") * hf[:, dff:]
        g.backward((dy.float() @ wt.float().t()).bfloat16().float())
        assert ((got.float() - hf.grad).norm() / hf.grad.norm()).item() < 4e-3


def test_gemm_rows_ok_says_which_shapes_fuse(lib):
    L = lib.load()
    assert L.mrmt3_gemm_rows_ok(65536, 512, 384, 384, 384) == 1
    assert L.mrmt3_gemm_rows_ok(65536, 1024, 512, 512, 512) == 1
    assert L.mrmt3_gemm_rows_ok(65536, 512, 96, 96, 96) == 0            # K off the 128 grid
    assert L.mrmt3_gemm_rows_ok(65536, 384, 512, 512, 512) == 0          # not the model width
    assert L.mrmt3_gemm_rows_ok(65536, 512, 384, 380, 384) == 0          # rows not 16-byte aligned
    assert L.mrmt3_gemm_rows_ok(4 << 20, 512, 384, 384, 384) == 0        # offsets beyond 2^31
    a = torch.zeros(64, 384, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        lib.gemm_nt_addnorm(a[:, :96], torch.zeros(512, 96, device="cuda", dtype=torch.bfloat16),
                            torch.zeros(64, 512, device="cuda"), torch.ones(512, device="cuda"), 1e-6)
