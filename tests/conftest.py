import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mr-mt3_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_abort_trace():
    """A process taken down by abort() inside HIP / RCCL leaves only Python frames behind (faulthandler); with the library
    loaded, its handler writes the NATIVE frames of the aborting thread first, then hands over to faulthandler
    (mrmt3_abort_trace_install).  MRMT3_ABORT_TRACE=0 switches it off; =<path> writes to a file instead of stderr."""
    where = os.environ.get("MRMT3_ABORT_TRACE", "")
    if where != "0":
        try:
            from mrmt3 import lib
            lib.abort_trace_install("" if where in ("", "1") else where)
        except Exception:                  # no library built: the tests that need it say so themselves
            pass
    yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "model_golden.npz"))


class _Knobs:
    """Dispatch / tuning switches for the duration of one test: `set(name, value)` puts MRMT3_<name> into the environment
    (for the Python readers: engine, trainer, ddp) AND overrides it in the C library (mrmt3_set_knob — the library reads
    its knobs from the environment once per process, so a later setenv alone would not reach it)."""

    def __init__(self, monkeypatch):
        self.mp, self.values = monkeypatch, {}

    def _apply(self):
        from mrmt3 import lib
        lib.reset_knobs()
        for k, v in self.values.items():
            try:
                lib.set_knob(k, int(v))
            except ValueError:
                pass                              # not an integer switch: a Python-side variable only

    def set(self, name, value):
        self.mp.setenv(name, str(value))
        self.values[name] = value
        self._apply()

    def unset(self, name):
        self.mp.delenv(name, raising=False)
        self.values.pop(name, None)
        self._apply()


@pytest.fixture
def knobs(monkeypatch):
    k = _Knobs(monkeypatch)
    yield k
    from mrmt3 import lib
    lib.reset_knobs()                              # (the environment is restored by monkeypatch after this)


def pytest_sessionfinish(session, exitstatus):
    """Orderly teardown while the interpreter and the HIP runtime are still whole: drain the device, then collect what the tests
    left behind NOW — captured hipGraphs, page-locked plan tables (`mrmt3_host_free` in `_PinnedTable.__del__`), RCCL communicators
    whose owner went away (`lib.Comm`'s finalizer) — instead of during interpreter shutdown, where the order in which torch, HIP
    and RCCL unload is not ours to choose.  (Added in round 5 while hunting two aborted suite runs; round 6 found their cause — the collector freeing such objects INSIDE a
    capture, profiles/r06_capture_abort_root_cause.txt — and fixed it in the trainer; an orderly exit is worth having either way.)"""
    import gc
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        else:
            gc.collect()
    except Exception:                      # teardown must never turn a green run red
        pass
