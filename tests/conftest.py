import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_l1(a, b):
    """mean|a-b| / mean|b| -- the metric BASELINE.json's north_star states (<= 1e-3)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).mean() / max(np.abs(b).mean(), 1e-30))


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o

    _o.build()
    return _o


def set_kernel(monkeypatch, kernel, on):
    """Take one specialised kernel (a key of config.KERNELS) out of / back into the dispatch for the duration of a test: the
    next-best kernel then serves the call (D3D_KERNELS_OFF)."""
    from deep3d_aerial_amd import config

    if kernel not in config.KERNELS:
        raise KeyError(kernel)
    names = {n for n in config.switches["D3D_KERNELS_OFF"].split(",") if n}
    names = (names - {kernel}) if on else (names | {kernel})
    monkeypatch.setitem(config.switches, "D3D_KERNELS_OFF", ",".join(sorted(names)))


def set_switch(monkeypatch, name, value):
    """One kernel-selection switch of deep3d_aerial_amd.config for the duration of a test (value None: its default).  The
    switches are read from the environment once, at import: a test changes the table, not os.environ."""
    from deep3d_aerial_amd import config

    monkeypatch.setitem(config.switches, name, config.SWITCHES[name][0] if value is None else str(value))
