import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def synth():
    import importlib
    return importlib.import_module("multicol-slam_amd.synth")


# MCS_CALL_TIMING=1: wall time of every C-ABI call the host classes make during the session (host buffers in and out, so H2D / D2H and the
# stream sync are included), printed per entry point at the end — the latency a live tracker sees, at the sizes the tests use (~3000 features).
class _TimedLib:
    def __init__(self, real, acc):
        self._real, self._acc = real, acc

    def __getattr__(self, name):
        import time
        fn = getattr(self._real, name)
        if not name.startswith("mcs_") or name in ("mcs_last_error",):
            return fn

        def timed(*a):
            t = time.perf_counter()
            r = fn(*a)
            self._acc.setdefault(name, []).append(time.perf_counter() - t)
            return r
        return timed


@pytest.fixture(scope="session", autouse=True)
def _call_timing():
    if os.environ.get("MCS_CALL_TIMING") != "1":
        yield
        return
    import importlib
    cap = importlib.import_module("multicol-slam_amd._capi")
    pkg = importlib.import_module("multicol-slam_amd")
    fe = importlib.import_module("multicol-slam_amd.frontend")
    acc = {}
    proxy = _TimedLib(cap.lib(), acc)
    for mod in (cap, pkg, fe):
        mod.lib = lambda proxy=proxy: proxy
    yield
    import numpy as np
    print("\nC-ABI call latency (ms): entry point, calls, median, min, max")
    for name in sorted(acc):
        v = 1e3 * np.array(acc[name])
        print("  %-34s %5d  %8.3f %8.3f %8.3f" % (name, len(v), np.median(v), v.min(), v.max()))
