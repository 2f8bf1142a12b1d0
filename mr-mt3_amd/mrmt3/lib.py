"""ctypes binding of libmrmt3_hip.so (include/mrmt3_hip.h).

The product path has NO CPU fallback: `load()` raises if the shared library is missing, and every
wrapper raises RuntimeError when the C ABI returns non-zero.  Tensors are passed as raw device
pointers; shapes/strides are explicit.  The stream is torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import torch

F32, BF16 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRMT3_TOOL_LIB") or os.path.join(_HERE, "libmrmt3_hip.so")   # env: A/B a variant build (tuning only)
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "mrmt3_hip.h")
CSRC_DIR = os.path.join(os.path.dirname(_HERE), "csrc")
_lib = None

vp, ci, cf, cu64, cu32, csz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t

_SIGS = {
    "mrmt3_version": (ci, []),
    "mrmt3_last_error": (C.c_char_p, []),
    "mrmt3_set_knob": (ci, [C.c_char_p, ci]),
    "mrmt3_reset_knobs": (ci, []),
    "mrmt3_logmel_fwd": (ci, [vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, ci, ci, vp, vp]),
    "mrmt3_logmel_crops_fwd": (ci, [vp, C.c_longlong, vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, ci, ci, vp, vp]),
    "mrmt3_gemm_nt": (ci, [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    "mrmt3_gemm_nt_workspace_bytes": (csz, [ci, ci, ci, ci]),
    "mrmt3_gemm_nt_ws": (ci, [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp, csz, vp]),
    "mrmt3_gemm_tn_workspace_bytes": (csz, [ci, ci, ci]),
    "mrmt3_gemm_tn": (ci, [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp, csz, vp]),
    "mrmt3_gemm_tn_splits": (ci, [ci, ci, ci]),
    "mrmt3_gemm_tn_partial": (ci, [vp, ci, vp, ci, ci, ci, ci, vp, csz, vp]),
    "mrmt3_tn_reduce_sites": (ci, [vp, ci, ci, vp]),
    "mrmt3_add_rmsnorm_fwd": (ci, [vp, vp, ci, vp, cf, vp, vp, ci, vp, ci, ci, cf, cu64, vp, cu32, cu32, ci, vp]),
    "mrmt3_add_rmsnorm_bwd_workspace_bytes": (csz, [ci, ci]),
    "mrmt3_add_rmsnorm_bwd_partial_rows": (ci, [ci]),
    "mrmt3_norm_dw_reduce": (ci, [vp, vp, vp, ci, ci, vp]),
    "mrmt3_add_rmsnorm_bwd": (ci, [vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, vp, ci, ci, cf, cu64, vp, cu32, cu32, ci, vp, csz, vp]),
    "mrmt3_attn_fwd": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_attn_bwd": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, ci, vp, vp, vp, ci, vp, ci, vp, ci,
                            ci, ci, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_geglu_fwd": (ci, [vp, vp, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_gemm_nt_geglu": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_gemm_rows_ok": (ci, [ci, ci, ci, ci, ci]),
    "mrmt3_gemm_nt_addnorm": (ci, [vp, ci, vp, ci, ci, ci, vp, vp, cf, vp, vp, vp, cf, cu64, vp, cu32, cu32, ci, vp]),
    "mrmt3_gemm_nt_normbwd_partial_rows": (ci, [ci]),
    "mrmt3_gemm_nt_normbwd": (ci, [vp, ci, vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, ci, vp, cf, cu64, vp, cu32, vp, csz, vp]),
    "mrmt3_gemm_nt_geglubwd": (ci, [vp, ci, vp, ci, vp, vp, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_tn_group_ok": (ci, [ci, ci, ci, ci, ci, ci]),
    "mrmt3_tn_group_plan": (ci, [vp, ci, vp, vp, C.c_size_t, vp]),
    "mrmt3_tn_group_run": (ci, [vp, vp, vp, vp]),
    "mrmt3_host_alloc": (vp, [C.c_size_t]),
    "mrmt3_host_free": (None, [vp]),
    "mrmt3_dispatch_counts": (ci, [vp, ci, ci]),
    "mrmt3_geglu_bwd": (ci, [vp, vp, vp, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_gemm_tn_f32": (ci, [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]),
    "mrmt3_attn_bwd_f32": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci,
                                cf, cu64, vp, cu32, vp]),
    "mrmt3_attn_fwd_bias": (ci, [vp, ci, vp, ci, vp, ci, vp, C.c_longlong, vp, ci, vp, ci, ci, ci, ci, ci, ci, cf, cu64, vp,
                                 cu32, vp]),
    "mrmt3_attn_bwd_bias": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, C.c_longlong, vp, ci, vp, ci, vp, ci, vp,
                                 ci, ci, ci, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_embed_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_embed_bwd_workspace_bytes": (csz, [ci, ci, ci]),
    "mrmt3_embed_bwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, cf, cu64, vp, cu32, vp, csz, vp]),
    "mrmt3_addpos_fwd": (ci, [vp, ci, vp, vp, ci, ci, ci, ci, cf, cu64, vp, cu32, vp]),
    "mrmt3_dropmask_cast": (ci, [vp, vp, ci, csz, cf, cu64, vp, cu32, vp]),
    "mrmt3_ce_count": (ci, [vp, ci, ci, ci, ci, vp, vp]),
    "mrmt3_ce_fwd_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, vp]),
    "mrmt3_lmhead_ce_fwd_bwd": (ci, [vp, ci, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, cf, vp, csz, ci, vp]),
    "mrmt3_adamw_step": (ci, [vp, vp, vp, vp, csz, vp, vp, cf, cf, cf, cf, cf, vp, vp]),
    "mrmt3_transpose": (ci, [vp, ci, vp, ci, ci, ci, vp]),
    "mrmt3_cast": (ci, [vp, ci, vp, ci, csz, vp]),
    "mrmt3_transpose_batched": (ci, [vp, vp, vp, vp, ci, ci, vp]),
    "mrmt3_decoder_create": (ci, [C.POINTER(vp), ci, ci, ci, ci, ci, ci, ci, ci, ci, cf]),
    "mrmt3_decoder_destroy": (None, [vp]),
    "mrmt3_decoder_begin": (ci, [vp, vp, vp, ci, ci, vp, ci, ci, ci, vp]),
    "mrmt3_decoder_set_prefix": (ci, [vp, vp, ci, vp]),
    "mrmt3_decoder_graph_captured": (ci, [vp]),
    "mrmt3_decoder_run": (ci, [vp, ci, vp]),
    "mrmt3_decoder_poll": (ci, [vp, vp, vp]),
    "mrmt3_comm_unique_id": (ci, [vp]),
    "mrmt3_comm_create": (ci, [vp, ci, ci, C.POINTER(vp)]),
    "mrmt3_comm_destroy": (ci, [vp]),
    "mrmt3_allreduce": (ci, [vp, vp, csz, ci, ci, vp]),
    "mrmt3_flag_signal": (ci, [vp, vp]),
    "mrmt3_flag_wait": (ci, [vp, vp, vp, ci, vp]),
    "mrmt3_stream_capture_status": (ci, [vp]),
    "mrmt3_stream_abandon_capture": (ci, [vp]),
    "mrmt3_runtime_error_pop": (ci, [C.c_char_p, ci]),
    "mrmt3_stream_create": (ci, [C.POINTER(vp), ci]),
    "mrmt3_stream_destroy": (ci, [vp]),
    "mrmt3_abort_trace_install": (ci, [C.c_char_p]),
}


def header_symbols(path: str = HEADER_PATH):
    """Names of every function the C header declares."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mrmt3_[a-z0-9_]+)\s*\(", txt)))


def build(verbose: bool = False, diag: bool = True) -> str:
    """Compile csrc/*.hip for gfx950 into libmrmt3_hip.so (hipcc cross-compiles without a GPU) and, with `diag`, the
    -DMRMT3_DIAG twin libmrmt3_hip_diag.so that profiles/tools load (never the product path)."""
    for target in (["all"], ["diag"]) if diag else (["all"],):
        r = subprocess.run(["make", "-C", CSRC_DIR, "-j8"] + target, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            print(r.stdout[-4000:], r.stderr[-4000:])
        if r.returncode != 0:
            raise RuntimeError("building libmrmt3_hip.so failed (make %s)" % target[0])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X kernels are the only implementation of this path "
            "(no CPU fallback). Run `python __graft_entry__.py` / `make -C mr-mt3_amd/csrc` first.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        if os.environ.get("MRMT3_TOOL_LIB") and not hasattr(lib, name):
            continue                       # tuning only: an older variant build under A/B may lack newer entry points
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if not os.environ.get("MRMT3_TOOL_LIB") and lib.mrmt3_version() < MIN_VERSION:
        raise RuntimeError(f"{LIB_PATH} is version {lib.mrmt3_version()}, this binding needs >= {MIN_VERSION}: "
                           "a stale build — run `make -C mr-mt3_amd/csrc`")
    _lib = lib
    return lib


MIN_VERSION = 110
COUNTER_NAMES = ("gemm_nt_tile", "gemm_nt8", "gemm_nt_geglu", "tn_group", "tn8", "tn_tile", "attn_fwd", "attn_bwd",
                 "attn_bwd_onepass", "attn_f32", "tn_f32", "gemm_nt_splitk", "gemm_nt_addnorm", "gemm_nt_normbwd",
                 "gemm_nt_geglubwd")


def set_knob(name: str, value: int):
    """Override a dispatch / tuning switch of the C library in-process (mrmt3_set_knob): `name` is the environment
    variable's name (MRMT3_ROWS_BM, MRMT3_GEMM8, ...), `value` the integer it would hold.  The library reads its knobs from
    the environment once per process; tests and tuning tools switch them through this call."""
    _check(load().mrmt3_set_knob(name.encode(), int(value)), "set_knob")


def reset_knobs():
    """Drop every override; each knob is re-read from the environment at its next use."""
    _check(load().mrmt3_reset_knobs(), "reset_knobs")


def dispatch_counts(reset: bool = False) -> dict:
    """Launches per kernel family since the process started / the last reset (mrmt3_dispatch_counts; diagnostics)."""
    arr = (C.c_ulonglong * len(COUNTER_NAMES))()
    n = load().mrmt3_dispatch_counts(arr, len(COUNTER_NAMES), int(reset))
    assert n == len(COUNTER_NAMES), "lib.COUNTER_NAMES is out of step with MRMT3_CNT_N"
    return {k: int(arr[i]) for i, k in enumerate(COUNTER_NAMES)}


# Optional per-launch timing used by bench.py's roofline leg: when PROFILE is a list, the wrappers of
# the heavy kernels bracket their launch with events on the current stream and append
# (family, algorithmic_work, unit, start_event, end_event).
PROFILE = None
PROFILE_BYTES = {}      # family -> algorithmic bytes (operands read once + output written once) while PROFILE is on


class _Timed:
    def __init__(self, family, work, unit, stream=None, executed=None):
        self.args = (family, work, unit)
        self.stream = stream
        self.executed = executed     # optional second price of the same launch (family "<name>@executed")

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(self.stream) if self.stream is not None else self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record(self.stream) if self.stream is not None else self.e1.record()
            PROFILE.append(self.args + (self.e0, self.e1))
            if self.executed is not None:
                PROFILE.append((self.args[0] + "@executed", self.executed, self.args[2], self.e0, self.e1))
        return False


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {load().mrmt3_last_error().decode()}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mrmt3 kernels need device tensors (no CPU fallback)")


# ---- thin wrappers (shape checks live in the C library) -----------------------------------------

def logmel(audio, tables, valid_frames=None, normalize=True, out_bf16=False):
    _dev(audio)
    B, n = audio.shape
    hop, n_mels = tables["hop"], tables["n_mels"]
    frames = -(-n // hop)
    out = torch.empty(B, frames, n_mels, device=audio.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    with _Timed("logmel", audio.numel() * 4 + out.numel() * out.element_size(), "B"):
        _check(load().mrmt3_logmel_fwd(_p(audio), B, n, hop, _p(tables["window"]), _p(tables["twiddle"]),
                                       _p(tables["fb_start"]), _p(tables["fb_cnt"]), _p(tables["fb_w"]), n_mels,
                                       tables["max_taps"], _p(valid_frames), int(normalize), int(out_bf16), _p(out),
                                       _stream()), "logmel_fwd")
    return out


def logmel_crops(audio, seg_start, n_samples, tables, valid_frames=None, normalize=True, out_bf16=False):
    """audio [total] f32 (one recording), seg_start [B] int64 sample offsets -> [B, ceil(n/hop), n_mels]."""
    _dev(audio, seg_start)
    assert audio.dim() == 1 and audio.dtype == torch.float32 and audio.is_contiguous()
    assert seg_start.dtype == torch.int64 and seg_start.is_contiguous()
    assert valid_frames is None or (valid_frames.dtype == torch.int32 and valid_frames.numel() == seg_start.numel())
    B = seg_start.numel()
    hop, n_mels = tables["hop"], tables["n_mels"]
    frames = -(-n_samples // hop)
    out = torch.empty(B, frames, n_mels, device=audio.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    with _Timed("logmel", B * n_samples * 4 + out.numel() * out.element_size(), "B"):
        _check(load().mrmt3_logmel_crops_fwd(_p(audio), audio.numel(), _p(seg_start), B, n_samples, hop,
                                             _p(tables["window"]), _p(tables["twiddle"]), _p(tables["fb_start"]),
                                             _p(tables["fb_cnt"]), _p(tables["fb_w"]), n_mels, tables["max_taps"],
                                             _p(valid_frames), int(normalize), int(out_bf16), _p(out), _stream()),
               "logmel_crops_fwd")
    return out


def gemm_nt(a, b, out=None, out_dtype=None, accumulate=False):
    """out[M,N] (+)= a[M,K] @ b[N,K]^T ; a/b are 2-D, last dim contiguous (row stride may exceed K)."""
    _dev(a, b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or a.dtype)
    assert out.stride(1) == 1
    fam = "gemm_nt_bf16" if a.dtype == torch.bfloat16 else "gemm_nt_f32"
    if PROFILE is not None:
        PROFILE_BYTES[fam] = PROFILE_BYTES.get(fam, 0.0) + (M * K + N * K) * a.element_size() + M * N * out.element_size()
    # short inputs with a long K (the encoder at 12 segments per GPU) run split over K through a scratch buffer
    key = (M, N, K, a.dtype)
    ws_bytes = _NT_WS.get(key)
    if ws_bytes is None:
        ws_bytes = _NT_WS[key] = int(load().mrmt3_gemm_nt_workspace_bytes(M, N, K, _dt(a)))
    with _Timed(fam, 2.0 * M * N * K, "FLOP"):
        if ws_bytes:
            ws = workspace(ws_bytes, a.device)
            _check(load().mrmt3_gemm_nt_ws(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, _dt(a),
                                           _dt(out), int(accumulate), _p(ws), ws.numel(), _stream()), "gemm_nt_ws")
        else:
            _check(load().mrmt3_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, _dt(a),
                                        _dt(out), int(accumulate), _stream()), "gemm_nt")
    return out


_NT_WS = {}          # (M, N, K, dtype) -> scratch bytes of the split-K form (0: the shape does not split)


_ws_cache = {}


_ws_retired = []


def workspace(nbytes: int, device, stream=None) -> torch.Tensor:
    # one scratch buffer per (device, stream): kernels on different streams may run concurrently
    key = (device.index if hasattr(device, "index") else 0,
           (torch.cuda.current_stream() if stream is None else stream).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # kernels already queued on that stream may still be using the old buffer, and the caching allocator
            # would hand its memory to the CURRENT stream: keep it (growth stops after the first steps)
            _ws_retired.append(buf)
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class TnBatch:
    """Weight-gradient GEMMs whose split-K slabs are summed in ONE launch instead of one per GEMM.

    `gemm_tn(..., defer=batch)` runs only the MFMA kernel, leaving the site's f32 slabs in a buffer that belongs to
    (batch, out address, shape), and queues the site; `flush()` runs `mrmt3_tn_reduce_sites` over everything queued,
    in queue order, on the current stream (which must already be ordered behind the GEMMs' stream).  Same summation
    order per element as the immediate form: bit-identical gradients.  The descriptor tables live on the device and
    are rebuilt only when the set of queued sites changes (never under graph capture after the eager warm-up)."""

    def __init__(self):
        self._slabs = {}       # (out address, M, N1, N2) -> slab tensor
        self._queue = []       # [(key, out, ldc, accumulate)]
        self._tables = {}
        # called before a flush that site() triggers by itself (the same gradient queued twice): the partial GEMMs of
        # the queued sites may have been launched on another stream (Engine.wgrad's side stream), which the CURRENT
        # stream — where the flush's reduce runs — has not joined yet.  The engine sets it to its join.
        self.before_early_flush = None

    def site(self, out, M, N1, N2, accumulate):
        key = (out.data_ptr(), M, N1, N2, out.stride(0), int(accumulate))
        if key in self._queue:
            # the same gradient twice before a flush: its slab buffer is still waiting to be summed
            if self.before_early_flush is not None:
                self.before_early_flush()
            self.flush()
        buf = self._slabs.get(key)
        if buf is None:
            buf = torch.empty(load().mrmt3_gemm_tn_workspace_bytes(M, N1, N2), device=out.device, dtype=torch.uint8)
            self._slabs[key] = buf
        self._queue.append(key)
        self._dev = out.device
        return buf

    def flush(self):
        if not self._queue:
            return
        import numpy as np
        keys = tuple(self._queue)
        tab = self._tables.get(keys)
        if tab is None:
            L = load()
            rec = np.zeros(len(keys), dtype=[("slabs", "<u8"), ("C", "<u8"), ("N1", "<i4"), ("N2", "<i4"), ("ldc", "<i4"),
                                             ("splits", "<i4"), ("acc", "<i4"), ("block0", "<i4"), ("p0", "<i4"), ("p1", "<i4")])
            blocks = 0
            for i, k in enumerate(keys):
                addr, M, N1, N2, ldc, acc = k
                rec[i] = (self._slabs[k].data_ptr(), addr, N1, N2, ldc, L.mrmt3_gemm_tn_splits(M, N1, N2), acc, blocks, 0, 0)
                blocks += -(-(N1 * N2) // 1024)
            tab = (torch.from_numpy(rec.view(np.uint8).copy()).to(self._dev), len(keys), blocks)
            self._tables[keys] = tab
        try:
            _check(load().mrmt3_tn_reduce_sites(_p(tab[0]), tab[1], tab[2], _stream()), "tn_reduce_sites")
        finally:
            self._queue.clear()


class _TnGSite(C.Structure):          # = mrmt3_tn_gsite
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("lda", "ldb", "ldc", "M", "N1", "N2", "accumulate", "pad")]


class _TnGInfo(C.Structure):          # = mrmt3_tn_group_info
    _fields_ = [(n, C.c_int32) for n in ("n_ctas", "n_items", "n_rtiles", "rounds")] + \
               [(n, C.c_uint64) for n in ("rtile_offset", "list_offset", "sync_offset", "table_bytes", "slab_bytes")]


class _PinnedTable:
    """Page-locked host bytes from the library's own allocator (torch's pinned-memory cache polls events when it
    allocates, which is illegal while a stream of the thread is capturing)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = load().mrmt3_host_alloc(self.nbytes)
        if not self.ptr:
            raise RuntimeError("mrmt3_host_alloc failed: " + load().mrmt3_last_error().decode())

    def __del__(self):
        try:
            if self.ptr:
                load().mrmt3_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = None


class TnGroup:
    """The weight-gradient GEMMs of a whole gradient bucket (or of the whole backward) in ONE MFMA launch + ONE reduce
    (mrmt3_tn_group_plan / mrmt3_tn_group_run).  `add()` only records the operands (and keeps them alive); `flush()`
    plans the items on the host — cached per set of (addresses, shapes) — and launches on the current stream; the table
    reaches the device through an async copy from page-locked memory (a memcpy node when the step is being captured:
    the replays re-send the same bytes).  Gradients are complete after flush().

    Nothing is allocated from the host while a stream is capturing: every eager plan leaves a spare page-locked table
    of its size behind, which the capture of the same step (same shapes, new addresses) picks up."""

    def __init__(self):
        self._sites = []
        self._plans = {}          # key -> dict(host, table, info, flops, slab, captured)
        self._spare = []          # page-locked tables ready for a plan made under capture
        self._slab = None
        self.last_info = None

    @staticmethod
    def ok(a, b, out):
        return (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and out.dtype == torch.float32 and
                a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1 and
                bool(load().mrmt3_tn_group_ok(a.shape[0], a.shape[1], b.shape[1], a.stride(0), b.stride(0), out.stride(0))))

    def add(self, a, b, out, accumulate=True):
        assert a.shape[0] == b.shape[0] and tuple(out.shape) == (a.shape[1], b.shape[1])
        if any(o.data_ptr() == out.data_ptr() for _, _, o, _ in self._sites):
            # the same gradient twice in one launch (a weight used twice per backward): its tiles would be reduced into
            # C by different workgroups without an order.  Finish the first use before the second is queued.
            self.flush()
        self._sites.append((a, b, out, int(bool(accumulate))))

    def __len__(self):
        return len(self._sites)

    def _host_table(self, nbytes, capturing):
        if not capturing:
            return _PinnedTable(nbytes)         # (the spares are for captures only: an eager plan must not use one up)
        fit = [i for i, h in enumerate(self._spare) if h.nbytes >= nbytes]
        if not fit:
            raise RuntimeError("TnGroup: no page-locked table of %d bytes was prepared before the capture "
                               "(run the step eagerly once with the same shapes first)" % nbytes)
        return self._spare.pop(min(fit, key=lambda i: self._spare[i].nbytes))

    def flush(self):
        if not self._sites:
            return
        try:
            self._flush()
        finally:
            self._sites.clear()        # also when planning / launching raised: the sites must not leak into the next flush

    def _flush(self):
        L = load()
        dev = self._sites[0][0].device
        key = tuple((a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), o.data_ptr(), o.stride(0), a.shape[0],
                     a.shape[1], b.shape[1], acc) for a, b, o, acc in self._sites)
        ent = self._plans.get(key)
        if ent is not None and ent["slab"] is not self._slab:
            ent = None
        if ent is None:
            capturing = torch.cuda.is_current_stream_capturing()
            n = len(self._sites)
            arr = (_TnGSite * n)()
            flops = 0.0
            for i, (a, b, o, acc) in enumerate(self._sites):
                arr[i] = _TnGSite(a.data_ptr(), b.data_ptr(), o.data_ptr(), a.stride(0), b.stride(0), o.stride(0),
                                  a.shape[0], a.shape[1], b.shape[1], acc, 0)
                flops += 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
            info = _TnGInfo()
            _check(L.mrmt3_tn_group_plan(arr, n, None, None, 0, C.byref(info)), "tn_group_plan (sizing)")
            if self._slab is None or self._slab.numel() < info.slab_bytes:
                self._slab = None
                self._slab = torch.empty(int(info.slab_bytes), device=dev, dtype=torch.uint8)
            if os.environ.get("MRMT3_TN_GROUP_DEBUG"):
                print("TnGroup plan: sites", n, "table", int(info.table_bytes), "capturing", capturing, "spares",
                      [h.nbytes for h in self._spare], "plans", len(self._plans), flush=True)
            host = self._host_table(int(info.table_bytes), capturing)
            _check(L.mrmt3_tn_group_plan(arr, n, _p(self._slab), C.c_void_p(host.ptr), host.nbytes, C.byref(info)),
                   "tn_group_plan")
            table = torch.empty(int(info.table_bytes), device=dev, dtype=torch.uint8)
            ent = dict(host=host, table=table, info=info, flops=flops, slab=self._slab, captured=capturing)
            if not capturing:
                # one spare per eager plan: a step with several joins (one per gradient bucket) needs as many tables
                # when it is captured
                self._spare.append(_PinnedTable(int(info.table_bytes)))
                if len(self._spare) > 64:
                    self._spare.pop(0)
                eager = [k for k, e in self._plans.items() if not e["captured"]]
                if len(eager) >= 128:                          # eager address churn (steady state reuses addresses)
                    torch.cuda.current_stream().synchronize()  # their table copies have been consumed
                    for k in eager:
                        del self._plans[k]
            self._plans[key] = ent
        with _Timed("gemm_tn_bf16", ent["flops"], "FLOP"):
            _check(L.mrmt3_tn_group_run(_p(ent["table"]), C.c_void_p(ent["host"].ptr), C.byref(ent["info"]), _stream()),
                   "tn_group_run")
        self.last_info = ent["info"]


def gemm_tn_f32(a, b, out, accumulate=False):
    """out[N1,N2] (+)= a[M,N1]^T @ b[M,N2], everything f32 (the fp32 training / parity path)."""
    _dev(a, b, out)
    M, N1 = a.shape
    N2 = b.shape[1]
    assert b.shape[0] == M and tuple(out.shape) == (N1, N2)
    assert a.dtype == b.dtype == out.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    with _Timed("gemm_tn_f32", 2.0 * M * N1 * N2, "FLOP"):
        _check(load().mrmt3_gemm_tn_f32(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N1, N2,
                                        int(accumulate), _stream()), "gemm_tn_f32")
    return out


def gemm_tn(a, b, out, accumulate=False, stream=None, defer=None):
    """out[N1,N2] (+)= a[M,N1]^T @ b[M,N2]  (bf16 in, f32 out).  `stream` (torch.cuda.Stream) launches there
    instead of on the current stream, without the cost of a stream context switch.  `defer=` a TnBatch: only the
    slabs are produced now, `out` is complete after the batch's flush()."""
    _dev(a, b, out)
    M, N1 = a.shape
    N2 = b.shape[1]
    assert b.shape[0] == M and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    lib = load()
    if defer is not None:
        slabs = defer.site(out, M, N1, N2, accumulate)
        sp = _stream() if stream is None else C.c_void_p(stream.cuda_stream)
        with _Timed("gemm_tn_bf16", 2.0 * M * N1 * N2, "FLOP", stream):
            _check(lib.mrmt3_gemm_tn_partial(_p(a), a.stride(0), _p(b), b.stride(0), M, N1, N2, _p(slabs),
                                             slabs.numel(), sp), "gemm_tn_partial")
        return out
    nbytes = lib.mrmt3_gemm_tn_workspace_bytes(M, N1, N2)
    ws = workspace(nbytes, a.device, stream)
    sp = _stream() if stream is None else C.c_void_p(stream.cuda_stream)
    with _Timed("gemm_tn_bf16", 2.0 * M * N1 * N2, "FLOP", stream):
        _check(lib.mrmt3_gemm_tn(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N1, N2,
                                 int(accumulate), _p(ws), ws.numel(), sp), "gemm_tn")
    return out


def add_rmsnorm_fwd(x0, y, w, eps, xn_dtype, write_x1=True, p=0.0, seed=0, stream_y=0, stream_out=0,
                    out_drop=False, x1=None, step=None):
    _dev(x0, y, w)
    rows, cols = x0.shape
    if x1 is None and write_x1:
        x1 = torch.empty_like(x0)
    xn = torch.empty(rows, cols, device=x0.device, dtype=xn_dtype)
    rstd = torch.empty(rows, device=x0.device, dtype=torch.float32)
    _check(load().mrmt3_add_rmsnorm_fwd(_p(x0), _p(y), _dt(y) if y is not None else F32, _p(w), eps, _p(x1), _p(xn),
                                        _dt(xn), _p(rstd), rows, cols, p, seed, _p(step), stream_y, stream_out,
                                        int(out_drop), _stream()), "add_rmsnorm_fwd")
    return (x1 if x1 is not None else x0), xn, rstd


class NormDwBatch:
    """Norm-weight gradients of one backward pass, summed in ONE launch instead of one per norm site.

    `add_rmsnorm_bwd(..., defer=batch)` leaves the site's per-workgroup partial rows in a workspace that belongs to
    (batch, dw vector, shape) and queues the site; `flush()` runs `mrmt3_norm_dw_reduce` over everything queued
    (same summation order as the immediate form, so the result is bit-identical).  The caller must flush before the
    gradients are read (the engine does so before a gradient bucket is sent and at the end of backward).  The address
    tables live on the device and are rebuilt only when the set of queued sites changes."""

    def __init__(self):
        self._ws = {}          # (dw address, rows, cols) -> workspace tensor
        self._queue = []       # [(key, dw tensor)]
        self._tables = {}      # tuple(keys) -> (ws addresses, dw addresses, partial rows) device tensors

    def site(self, dw, rows, cols, n_part=None):
        """`n_part`: partial rows the producing kernel leaves (default: the stand-alone norm backward's count)."""
        if n_part is None:
            n_part = load().mrmt3_add_rmsnorm_bwd_partial_rows(rows)
        key = (dw.data_ptr(), rows, cols, n_part)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(load().mrmt3_add_rmsnorm_bwd_workspace_bytes(rows, cols), device=dw.device, dtype=torch.uint8)
            self._ws[key] = ws
        self._queue.append((key, dw))
        return ws

    def flush(self):
        if not self._queue:
            return
        for cols in sorted({k[2] for k, _ in self._queue}):
            keys = tuple(k for k, _ in self._queue if k[2] == cols)
            tab = self._tables.get(keys)
            dev = self._queue[0][1].device
            if tab is None:
                tab = (torch.tensor([self._ws[k].data_ptr() for k in keys], dtype=torch.int64, device=dev),
                       torch.tensor([k[0] for k in keys], dtype=torch.int64, device=dev),
                       torch.tensor([k[3] for k in keys], dtype=torch.int32, device=dev))
                self._tables[keys] = tab
            _check(load().mrmt3_norm_dw_reduce(_p(tab[0]), _p(tab[1]), _p(tab[2]), len(keys), cols, _stream()),
                   "norm_dw_reduce")
        self._queue.clear()


def add_rmsnorm_bwd(dxn, dres, x1, rstd, w, dw, want_dy=True, p=0.0, seed=0, stream_y=0, stream_out=0,
                    out_drop=False, dx1=None, dx1_dtype=torch.float32, defer=None, step=None):
    """`dres` and the returned dx1 may be f32 or bf16 (the bf16 engine's residual-gradient stream); `dx1=`
    reuses a buffer (in place when it is `dres` itself).  `defer=` a NormDwBatch: dw is produced by its flush()."""
    _dev(dxn, x1, rstd, w)
    rows, cols = x1.shape
    if dx1 is None:
        dx1 = torch.empty(rows, cols, device=x1.device, dtype=dx1_dtype)
    dy = torch.empty(rows, cols, device=x1.device, dtype=torch.bfloat16) if want_dy else None
    if dw is None:
        ws = None
    elif defer is not None:
        ws, dw = defer.site(dw, rows, cols), None
    else:
        ws = workspace(load().mrmt3_add_rmsnorm_bwd_workspace_bytes(rows, cols), x1.device)
    _check(load().mrmt3_add_rmsnorm_bwd(_p(dxn), _dt(dxn), _p(dres), _dt(dres) if dres is not None else F32, _p(x1),
                                        _p(rstd), _p(w), _p(dx1), _dt(dx1), _p(dy), _p(dw), rows, cols, p, seed,
                                        _p(step), stream_y, stream_out, int(out_drop), _p(ws), ws.numel() if ws is not None else 0,
                                        _stream()),
           "add_rmsnorm_bwd")
    return dx1, dy


def _attn_executed_pairs(Lq, Lk, causal):
    """(query, key) pairs the flash kernels visit: 128-query tiles x 64-key tiles, causal tiles above the diagonal skipped."""
    if not causal:
        return float(Lq) * Lk
    n = 0
    for q0 in range(0, Lq, 128):
        n_kv = min(-(-Lk // 64), min(q0 + 127, Lq - 1) // 64 + 1)
        n += min(128, Lq - q0) * min(Lk, n_kv * 64)
    return float(n)


def attn_fwd(q, k, v, B, H, Lq, Lk, causal, p=0.0, seed=0, stream_id=0, want_lse=True, step=None, want_lo=None, out=None,
             out_lo=None):
    """q: [B*Lq, ldq-view], k/v: [B*Lk, ld-view] 2-D views whose column 0 is head 0 / dim 0.
    want_lo (True / False instead of None): returns (o, lse, o_lo) with o_lo = bf16(O - bf16(O)) for the backward's
    delta when True (and the kernel is the bf16 one), else None.  out / out_lo: caller-owned [B*Lq, H*64] views of the same
    row stride (a multiple of 4 elements; rows that are not 16-byte aligned take the kernel's 8-byte store path)."""
    _dev(q, k, v)
    o = torch.empty(B * Lq, H * 64, device=q.device, dtype=q.dtype) if out is None else out
    assert o.shape == (B * Lq, H * 64) and o.dtype == q.dtype and o.stride(1) == 1
    if out_lo is not None:
        assert want_lo and out_lo.shape == o.shape and out_lo.stride(0) == o.stride(0) and out_lo.dtype == o.dtype
    o_lo = (out_lo if out_lo is not None else torch.empty_like(o)) if want_lo and q.dtype == torch.bfloat16 else None
    assert o_lo is None or o_lo.stride(0) == o.stride(0)
    lse = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32) if want_lse else None
    # algorithmic FLOPs count the full (unskipped) square, as the reference computes it
    with _Timed("attn_fwd", 4.0 * B * H * Lq * Lk * 64, "FLOP", executed=4.0 * B * H * 64 * _attn_executed_pairs(Lq, Lk, causal)):
        _check(load().mrmt3_attn_fwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                     _p(o_lo), _p(lse), B, H, Lq, Lk, int(causal), _dt(q), p, seed, _p(step), stream_id,
                                     _stream()),
               "attn_fwd")
    if want_lo is not None:
        return o, lse, o_lo
    return o, lse


def attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, Lk, causal, p=0.0, seed=0, stream_id=0, step=None, o_lo=None):
    _dev(q, k, v, o, d_o, lse, dq, dk, dv, o_lo)
    assert o_lo is None or (o_lo.shape == o.shape and o_lo.stride(0) == o.stride(0))
    delta = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    if q.dtype == torch.float32:                     # exact-f32 path (fp32 training / parity)
        assert all(t.dtype == torch.float32 for t in (k, v, o, d_o, dq, dk, dv))
        _check(load().mrmt3_attn_bwd_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                         _p(d_o), d_o.stride(0), _p(lse), _p(delta), _p(dq), dq.stride(0), _p(dk),
                                         dk.stride(0), _p(dv), dv.stride(0), B, H, Lq, Lk, int(causal), p, seed,
                                         _p(step), stream_id, _stream()), "attn_bwd_f32")
        return dq, dk, dv
    with _Timed("attn_bwd", 8.0 * B * H * Lq * Lk * 64, "FLOP", executed=8.0 * B * H * 64 * _attn_executed_pairs(Lq, Lk, causal)):
        _check(load().mrmt3_attn_bwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                     _p(o_lo), _p(d_o), d_o.stride(0), _p(lse), _p(delta), _p(dq), dq.stride(0), _p(dk),
                                     dk.stride(0), _p(dv), dv.stride(0), B, H, Lq, Lk, int(causal), p, seed,
                                     _p(step), stream_id, _stream()), "attn_bwd")
    return dq, dk, dv


def _bias_stride(bias, B, H, Lq, Lk):
    assert bias.dtype == torch.float32 and bias.is_contiguous()
    if tuple(bias.shape) == (H, Lq, Lk):
        return 0
    assert tuple(bias.shape) == (B, H, Lq, Lk), "bias is [H, Lq, Lk] (shared by the batch) or [B, H, Lq, Lk]"
    return H * Lq * Lk


def attn_fwd_bias(q, k, v, bias, B, H, Lq, Lk, causal, p=0.0, seed=0, stream_id=0, step=None):
    """softmax(q k^T + bias [+ causal]) v — HF T5Attention with its additive position bias (models/t5.py:636-648).  The
    general kernel (f32 arithmetic on f32 / bf16 operands), not the MFMA path: MR-MT3's own bias is zero and its step
    calls attn_fwd.  Returns (o, lse)."""
    _dev(q, k, v, bias)
    o = torch.empty(B * Lq, H * 64, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    bs = 0 if bias is None else _bias_stride(bias, B, H, Lq, Lk)
    _check(load().mrmt3_attn_fwd_bias(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(bias), bs, _p(o),
                                      o.stride(0), _p(lse), B, H, Lq, Lk, int(causal), _dt(q), p, seed, _p(step), stream_id,
                                      _stream()), "attn_fwd_bias")
    return o, lse


def attn_bwd_bias(q, k, v, o, d_o, lse, bias, B, H, Lq, Lk, causal, p=0.0, seed=0, stream_id=0, step=None, want_dbias=True):
    """Backward of attn_fwd_bias: returns (dq, dk, dv, dbias); dbias has the bias's shape (summed over the batch when the
    bias is shared) or is None."""
    _dev(q, k, v, o, d_o, lse, bias)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
    bs = 0 if bias is None else _bias_stride(bias, B, H, Lq, Lk)
    dbias = torch.empty_like(bias) if (want_dbias and bias is not None) else None
    _check(load().mrmt3_attn_bwd_bias(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                      _p(d_o), d_o.stride(0), _p(lse), _p(delta), _p(bias), bs, _p(dq), dq.stride(0), _p(dk),
                                      dk.stride(0), _p(dv), dv.stride(0), _p(dbias), B, H, Lq, Lk, int(causal), _dt(q), p,
                                      seed, _p(step), stream_id, _stream()), "attn_bwd_bias")
    return dq, dk, dv, dbias


def geglu_fwd(h, p=0.0, seed=0, stream_id=0, step=None):
    _dev(h)
    rows, two = h.shape
    g = torch.empty(rows, two // 2, device=h.device, dtype=h.dtype)
    _check(load().mrmt3_geglu_fwd(_p(h), _p(g), rows, two // 2, _dt(h), p, seed, _p(step), stream_id, _stream()),
           "geglu_fwd")
    return g


def gemm_nt_geglu(x, wi, p=0.0, seed=0, stream_id=0, step=None):
    """h = x . wi^T and g = dropout(gelu_new(h0) * h1) in one launch (bf16); returns (h, g), bit-identical to
    gemm_nt + geglu_fwd."""
    assert x.dtype == torch.bfloat16 and wi.dtype == torch.bfloat16 and x.stride(-1) == 1 and wi.stride(-1) == 1
    rows, K = x.shape
    two = wi.shape[0]
    h = torch.empty(rows, two, device=x.device, dtype=torch.bfloat16)
    g = torch.empty(rows, two // 2, device=x.device, dtype=torch.bfloat16)
    with _Timed("gemm_nt_geglu_bf16", 2.0 * rows * two * K, "FLOP"):      # (its own family: GEMM + gated-GELU epilogue)
        _check(load().mrmt3_gemm_nt_geglu(_p(x), x.stride(0), _p(wi), wi.stride(0), _p(h), two, _p(g), two // 2, rows,
                                          two // 2, K, p, seed, _p(step), stream_id, _stream()), "gemm_nt_geglu")
    return h, g


def gemm_rows_ok(a, w, N=512):
    """True when the fused projection + row kernels (gemm_rows.hip) take a [M, K] x [N, K]^T product of these operands."""
    return (a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2 and
            a.stride(1) == 1 and w.stride(1) == 1 and w.shape[0] == N and a.shape[1] == w.shape[1] and
            bool(load().mrmt3_gemm_rows_ok(a.shape[0], N, a.shape[1], a.stride(0), w.stride(0))))


def _rows_bytes(M, N, K, row_bytes):
    """algorithmic bytes of a fused launch: operands once + `row_bytes` per output element of the row operands"""
    return (M * K + N * K) * 2 + M * N * row_bytes


def gemm_nt_addnorm(a, w_proj, x0, w_norm, eps, write_x1=True, p=0.0, seed=0, stream_y=0, stream_out=0, out_drop=False,
                    x1=None, step=None):
    """(x1, xn, rstd) = add_rmsnorm_fwd(x0, gemm_nt(a, w_proj), w_norm, ...) in ONE launch (bf16; same bits)."""
    _dev(a, w_proj, x0, w_norm)
    rows, K = a.shape
    assert x0.shape == (rows, 512) and x0.dtype == torch.float32 and x0.is_contiguous()
    if x1 is None and write_x1:
        x1 = torch.empty_like(x0)
    xn = torch.empty(rows, 512, device=a.device, dtype=torch.bfloat16)
    rstd = torch.empty(rows, device=a.device, dtype=torch.float32)
    if PROFILE is not None:
        PROFILE_BYTES["gemm_nt_addnorm_bf16"] = PROFILE_BYTES.get("gemm_nt_addnorm_bf16", 0.0) + _rows_bytes(rows, 512, K, 10)
    with _Timed("gemm_nt_addnorm_bf16", 2.0 * rows * 512 * K, "FLOP"):
        _check(load().mrmt3_gemm_nt_addnorm(_p(a), a.stride(0), _p(w_proj), w_proj.stride(0), rows, K, _p(x0), _p(w_norm),
                                            eps, _p(x1), _p(xn), _p(rstd), p, seed, _p(step), stream_y, stream_out,
                                            int(out_drop), _stream()), "gemm_nt_addnorm")
    return (x1 if x1 is not None else x0), xn, rstd


def gemm_nt_normbwd(a, wt, dres, x1, rstd, w_norm, dw, want_dy=True, p=0.0, seed=0, stream_y=0, dx1=None,
                    dx1_dtype=torch.float32, defer=None, step=None):
    """(dx1, dy) = add_rmsnorm_bwd(gemm_nt(a, wt), dres, x1, rstd, w_norm, dw, ...) in ONE launch (bf16 product)."""
    _dev(a, wt, dres, x1, rstd, w_norm)
    rows, K = a.shape
    assert x1.shape == (rows, 512) and dres is not None and dres.shape == (rows, 512)
    if dx1 is None:
        dx1 = torch.empty(rows, 512, device=a.device, dtype=dx1_dtype)
    dy = torch.empty(rows, 512, device=a.device, dtype=torch.bfloat16) if want_dy else None
    L = load()
    n_part = L.mrmt3_gemm_nt_normbwd_partial_rows(rows)
    reduce_now = None
    if dw is None:
        ws = None
    elif defer is not None:
        ws = defer.site(dw, rows, 512, n_part)
    else:
        ws = workspace(L.mrmt3_add_rmsnorm_bwd_workspace_bytes(rows, 512), a.device)
        reduce_now = dw
    if PROFILE is not None:
        PROFILE_BYTES["gemm_nt_normbwd_bf16"] = PROFILE_BYTES.get("gemm_nt_normbwd_bf16", 0.0) + _rows_bytes(
            rows, 512, K, 4 + dres.element_size() + dx1.element_size() + (2 if want_dy else 0))
    with _Timed("gemm_nt_normbwd_bf16", 2.0 * rows * 512 * K, "FLOP"):
        _check(L.mrmt3_gemm_nt_normbwd(_p(a), a.stride(0), _p(wt), wt.stride(0), rows, K, _p(dres), _dt(dres), _p(x1),
                                       _p(rstd), _p(w_norm), _p(dx1), _dt(dx1), _p(dy), p, seed, _p(step), stream_y,
                                       _p(ws), ws.numel() if ws is not None else 0, _stream()), "gemm_nt_normbwd")
    if reduce_now is not None:
        # the one-site address table of the reduce, cached per (workspace, dw, partial rows): built by a synchronous
        # host-to-device copy, which a hipGraph capture of the step does not allow — the eager warm-up steps leave it here
        key = (ws.data_ptr(), dw.data_ptr(), n_part, a.device.index)
        tab = _NORMBWD_TABLES.get(key)
        if tab is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("gemm_nt_normbwd: the norm-weight reduce table of this site was not prepared before the "
                                   "capture (run the step eagerly once with the same buffers, or pass defer=NormDwBatch)")
            dev = a.device
            tab = _NORMBWD_TABLES[key] = (torch.tensor([ws.data_ptr()], dtype=torch.int64, device=dev),
                                          torch.tensor([dw.data_ptr()], dtype=torch.int64, device=dev),
                                          torch.tensor([n_part], dtype=torch.int32, device=dev))
            if len(_NORMBWD_TABLES) > 256:
                _NORMBWD_TABLES.pop(next(iter(_NORMBWD_TABLES)))
        _check(L.mrmt3_norm_dw_reduce(_p(tab[0]), _p(tab[1]), _p(tab[2]), 1, 512, _stream()), "norm_dw_reduce")
    return dx1, dy


_NORMBWD_TABLES = {}


def gemm_nt_geglubwd(dy, wt, h, p=0.0, seed=0, stream_id=0, step=None):
    """dh = geglu_bwd(h, gemm_nt(dy, wt)) in ONE launch (wt = wo^T [dff, d]; bf16; same bits)."""
    _dev(dy, wt, h)
    rows, K = dy.shape
    dff = wt.shape[0]
    assert h.shape == (rows, 2 * dff) and h.dtype == torch.bfloat16 and h.is_contiguous()
    dh = torch.empty_like(h)
    if PROFILE is not None:
        PROFILE_BYTES["gemm_nt_geglubwd_bf16"] = PROFILE_BYTES.get("gemm_nt_geglubwd_bf16", 0.0) + _rows_bytes(rows, dff, K, 8)
    with _Timed("gemm_nt_geglubwd_bf16", 2.0 * rows * dff * K, "FLOP"):
        _check(load().mrmt3_gemm_nt_geglubwd(_p(dy), dy.stride(0), _p(wt), wt.stride(0), _p(h), _p(dh), rows, dff, K, p,
                                             seed, _p(step), stream_id, _stream()), "gemm_nt_geglubwd")
    return dh


def geglu_bwd(h, dg, p=0.0, seed=0, stream_id=0, step=None):
    _dev(h, dg)
    rows, two = h.shape
    dh = torch.empty_like(h)
    assert dg.dtype == h.dtype
    _check(load().mrmt3_geglu_bwd(_p(h), _p(dg), _p(dh), rows, two // 2, _dt(h), p, seed, _p(step), stream_id, _stream()),
           "geglu_bwd")
    return dh


def embed_fwd(ids, table, pos, seq_len, shift, start_id=0, pad_id=0, pos_offset=0, p=0.0, seed=0, stream_id=0,
              step=None):
    _dev(ids, table, pos)
    rows = ids.numel()
    V, d = table.shape
    x = torch.empty(rows, d, device=table.device, dtype=torch.float32)
    _check(load().mrmt3_embed_fwd(_p(ids), _p(table), _p(pos), _p(x), rows, seq_len, d, V, int(shift), start_id,
                                  pad_id, pos_offset, p, seed, _p(step), stream_id, _stream()), "embed_fwd")
    return x


def embed_bwd(ids, dx, dtable, seq_len, shift, start_id=0, pad_id=0, p=0.0, seed=0, stream_id=0, step=None):
    _dev(ids, dx, dtable)
    rows = ids.numel()
    V, d = dtable.shape
    ws = workspace(load().mrmt3_embed_bwd_workspace_bytes(rows, V, d), dx.device)
    _check(load().mrmt3_embed_bwd(_p(ids), _p(dx), _p(dtable), rows, seq_len, d, V, int(shift), start_id, pad_id, p,
                                  seed, _p(step), stream_id, _p(ws), ws.numel(), _stream()), "embed_bwd")


def addpos_fwd(src, pos, seq_len, pos_offset=0, p=0.0, seed=0, stream_id=0, step=None):
    _dev(src, pos)
    rows, d = src.shape
    x = torch.empty(rows, d, device=src.device, dtype=torch.float32)
    _check(load().mrmt3_addpos_fwd(_p(src), _dt(src), _p(pos), _p(x), rows, seq_len, d, pos_offset, p, seed,
                                   _p(step), stream_id, _stream()), "addpos_fwd")
    return x


def dropmask_cast(dx, p=0.0, seed=0, stream_id=0, step=None, out_dtype=torch.bfloat16):
    _dev(dx)
    out = torch.empty(dx.shape, device=dx.device, dtype=out_dtype)
    _check(load().mrmt3_dropmask_cast(_p(dx), _p(out), _dt(out), dx.numel(), p, seed, _p(step), stream_id, _stream()),
           "dropmask_cast")
    return out


def cross_entropy(logits, targets, want_grad=True, grad_dtype=torch.bfloat16, weighted=False, inst_lo=1135,
                  inst_hi=1262, grad_scale=1.0):
    """Returns (loss_dev[1] f32 tensor, dlogits or None).  No host sync."""
    _dev(logits, targets)
    rows, V = logits.shape
    acc = torch.zeros(2, device=logits.device, dtype=torch.float64)  # [loss (double accumulator), denom (f32 in its first 4 bytes)]
    den = C.c_void_p(acc.data_ptr() + 8)
    lib = load()
    _check(lib.mrmt3_ce_count(_p(targets), rows, int(weighted), inst_lo, inst_hi, den, _stream()), "ce_count")
    dl = torch.empty(rows, V, device=logits.device, dtype=grad_dtype) if want_grad else None
    _check(lib.mrmt3_ce_fwd_bwd(_p(logits), _p(targets), den, _p(acc), _p(dl),
                                _dt(dl) if dl is not None else F32, rows, V, int(weighted), inst_lo, inst_hi,
                                grad_scale, _stream()), "ce_fwd_bwd")
    return acc[0:1].float(), dl


def ce_chunk_rows() -> int:
    """MRMT3_CE_CHUNK (rows of logits per lm_head + CE chunk), validated: an integer, at least 1024."""
    raw = os.environ.get("MRMT3_CE_CHUNK", "65536")
    try:
        n = int(raw)
    except ValueError:
        raise ValueError(f"MRMT3_CE_CHUNK={raw!r} is not an integer") from None
    if n < 1024:
        raise ValueError(f"MRMT3_CE_CHUNK={n}: a chunk is at least 1024 rows")
    return n


def lmhead_cross_entropy(dec, w, targets, want_grad=True, grad_dtype=torch.bfloat16, weighted=False, inst_lo=1135,
                         inst_hi=1262, grad_scale=1.0, chunk_rows=None):
    """lm_head + CE fused over row chunks (mrmt3_lmhead_ce_fwd_bwd): dec [rows, d] bf16, w [V, d] bf16 ->
    (loss_dev[1] f32, dlogits [rows, V] or None).  The f32 logits only ever exist one chunk at a time, in a workspace."""
    _dev(dec, w, targets)
    rows, d = dec.shape
    V = w.shape[0]
    assert dec.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and dec.stride(1) == 1 and w.stride(1) == 1
    if chunk_rows is None:
        # 65 536 rows = 403 MB of f32 logits in the workspace: the 64-segment step in ONE chunk (25.41 -> 25.32 ms against
        # four chunks of 16 384, same box); larger batches still go chunk by chunk (MRMT3_CE_CHUNK)
        chunk_rows = ce_chunk_rows()
    acc = torch.zeros(2, device=dec.device, dtype=torch.float64)  # [loss (double accumulator), denom (f32 in its first 4 bytes)]
    den = C.c_void_p(acc.data_ptr() + 8)
    lib = load()
    _check(lib.mrmt3_ce_count(_p(targets), rows, int(weighted), inst_lo, inst_hi, den, _stream()), "ce_count")
    dl = torch.empty(rows, V, device=dec.device, dtype=grad_dtype) if want_grad else None
    ws = workspace(min(rows, chunk_rows) * V * 4, dec.device)
    _check(lib.mrmt3_lmhead_ce_fwd_bwd(_p(dec), dec.stride(0), _p(w), w.stride(0), _p(targets),
                                       den, _p(acc), _p(dl), _dt(dl) if dl is not None else F32,
                                       rows, V, d, int(weighted), inst_lo, inst_hi, grad_scale, _p(ws), ws.numel(),
                                       chunk_rows, _stream()), "lmhead_ce_fwd_bwd")
    return acc[0:1].float(), dl


def adamw_step(p, g, m, v, lr_dev, step_dev, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0,
               shadow=None):
    _dev(p, g, m, v, lr_dev, step_dev)
    _check(load().mrmt3_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_dev), _p(step_dev), beta1, beta2,
                                   eps, weight_decay, grad_scale, _p(shadow), _stream()), "adamw_step")


def transpose(src, out):
    _dev(src, out)
    rows, cols = src.shape
    _check(load().mrmt3_transpose(_p(src), _dt(src), _p(out), _dt(out), rows, cols, _stream()), "transpose")
    return out


def cast(src, out):
    _dev(src, out)
    _check(load().mrmt3_cast(_p(src), _dt(src), _p(out), _dt(out), src.numel(), _stream()), "cast")
    return out


def transpose_batched(src_flat, dst_flat, desc_table, tile_start, n_mats, total_tiles):
    _dev(src_flat, dst_flat, desc_table, tile_start)
    _check(load().mrmt3_transpose_batched(_p(src_flat), _p(dst_flat), _p(desc_table), _p(tile_start), n_mats,
                                          total_tiles, _stream()), "transpose_batched")


COMM_ID_BYTES = 128


class Comm:
    """An RCCL communicator behind the C ABI (`mrmt3_comm_*`, `mrmt3_allreduce`): the gradient exchange of the data-parallel
    step without torch.distributed in the data path (mrmt3/ddp.py, MRMT3_DDP_NATIVE=1).  `uid` = the 128 bytes rank 0 got from
    `Comm.unique_id()`, handed to the other ranks by whatever channel the host has (ddp.py: one broadcast_object_list).
    Creation blocks until every rank has created its end."""

    def __init__(self, uid: bytes, rank: int, world: int):
        assert len(uid) == COMM_ID_BYTES
        self.rank, self.world = rank, world
        self._h = vp()
        buf = C.create_string_buffer(uid, COMM_ID_BYTES)
        _check(load().mrmt3_comm_create(buf, rank, world, C.byref(self._h)), "comm_create")
        # a communicator nobody closed is still destroyed (ncclCommDestroy) when the object goes away or the interpreter
        # exits — before torch's own process group is torn down at exit
        import weakref
        self._fin = weakref.finalize(self, Comm._destroy, self._h)

    @staticmethod
    def _destroy(h):
        if h:
            load().mrmt3_comm_destroy(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(load().mrmt3_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    def allreduce(self, t: torch.Tensor, average: bool = False, stream=None):
        """In place, asynchronous on `stream` (default: the current stream): t = sum (or mean) of t over the ranks."""
        _dev(t)
        assert t.is_contiguous()
        s = C.c_void_p((stream or torch.cuda.current_stream(t.device)).cuda_stream)
        _check(load().mrmt3_allreduce(self._h, _p(t), t.numel(), _dt(t), int(average), s), "allreduce")
        return t

    def close(self):
        if self._h:
            h, self._h = self._h, vp()
            self._fin.detach()
            _check(load().mrmt3_comm_destroy(h), "comm_destroy")


def flag_signal(flag: torch.Tensor, stream=None):
    """flag[0] += 1 behind everything enqueued so far on `stream` (default: the current one) — mrmt3_flag_signal."""
    _dev(flag)
    assert flag.dtype == torch.int32 and flag.numel() >= 1
    s = C.c_void_p((stream or torch.cuda.current_stream(flag.device)).cuda_stream)
    _check(load().mrmt3_flag_signal(_p(flag), s), "flag_signal")


def flag_wait(flag: torch.Tensor, seen: torch.Tensor, err: torch.Tensor, timeout_ms: int = 5000, stream=None):
    """`stream` waits until flag[0] >= seen[0] + 1, then seen[0] += 1; err[0] = 1 after `timeout_ms` without the signal."""
    _dev(flag, seen, err)
    assert flag.dtype == seen.dtype == err.dtype == torch.int32
    s = C.c_void_p((stream or torch.cuda.current_stream(flag.device)).cuda_stream)
    _check(load().mrmt3_flag_wait(_p(flag), _p(seen), _p(err), int(timeout_ms), s), "flag_wait")


# ---- capture hygiene (include/mrmt3_hip.h: "capture hygiene") ------------------------------------------------------------
_CAPTURE_STATUS = {0: "none", 1: "active", 2: "invalidated"}


def stream_capture_status(stream) -> str:
    """'none' / 'active' / 'invalidated' for a torch stream (or a raw handle)."""
    h = stream if isinstance(stream, int) else stream.cuda_stream
    r = load().mrmt3_stream_capture_status(C.c_void_p(h))
    return _CAPTURE_STATUS.get(r, "error(%d)" % r)


def stream_abandon_capture(stream) -> str:
    """End a capture that `stream` is still in (destroying its graph); returns the status before the call."""
    h = stream if isinstance(stream, int) else stream.cuda_stream
    r = load().mrmt3_stream_abandon_capture(C.c_void_p(h))
    return _CAPTURE_STATUS.get(r, "error(%d)" % r)


def runtime_error_pop() -> str:
    """Name of the calling thread's pending HIP runtime error ('' = none); the slot is empty afterwards."""
    buf = C.create_string_buffer(96)
    load().mrmt3_runtime_error_pop(buf, 96)
    return buf.value.decode()


def abort_trace_install(path: str = "") -> None:
    """Opt-in: native frames of the faulting thread on SIGABRT / SIGSEGV (to `path`, default stderr)."""
    _check(load().mrmt3_abort_trace_install(path.encode() if path else None), "abort_trace_install")


class OwnedStream:
    """A HIP stream of our own (mrmt3_stream_create), wrapped for torch as an ExternalStream: NOT from torch's pool of 32 streams
    per priority, which hands a stream that a failed capture left invalidated back out a few dozen `Stream()` calls later.
    `.stream` is the torch handle; close() destroys the HIP stream (call it only when nothing on it is pending)."""

    def __init__(self, device=None, priority: int = 0):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = vp()
        with torch.cuda.device(dev):
            _check(load().mrmt3_stream_create(C.byref(h), int(priority)), "stream_create")
        self.handle, self.device = h.value, dev
        self.stream = torch.cuda.ExternalStream(self.handle, device=dev)

    def close(self):
        h, self.handle = self.handle, None
        if h and _lib is not None:
            self.stream = None
            _lib.mrmt3_stream_destroy(vp(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
