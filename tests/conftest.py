"""pytest configuration: the `gpu` marker, the CPU oracle build and shared scan fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the suite under `-x`: per-kernel parity first, the long / multi-pipeline / whole-sequence cases last, so that
# one late failure can never hide the per-kernel evidence again (round 5: a stopwatch assert in test_gpu_long.py stopped
# the driver's run before test_gpu_parity.py had started).  Files not named here keep their alphabetical place between.
_ORDER_FIRST = ["test_abi", "test_gpu_parity", "test_gpu_phases", "test_gpu_dist", "test_ref_golden", "test_golden",
                "test_gpu_gl_golden", "test_gpu_cpp", "test_ref_shells", "test_gpu_adapter"]
_ORDER_LAST = ["test_gpu_long"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _ORDER_FIRST:
            return (0, _ORDER_FIRST.index(mod))
        if mod in _ORDER_LAST:
            return (2, _ORDER_LAST.index(mod))
        return (1, 0)
    items.sort(key=key)  # stable: the order inside a file is the file's


@pytest.fixture(scope="session")
def oracle_lib():
    """TEST INFRASTRUCTURE: the CPU restatement of the reference path (oracle/), built with gcc."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


_SCAN_CACHE = {}


def get_scan(k, width, semantics=True, height=64):
    """Deterministic synthetic scan k for a `width`-column image (cached per session)."""
    from semantic_suma_amd import synth
    key = (k, width, semantics, height)
    if key not in _SCAN_CACHE:
        # ~W azimuth steps x H beams, like a real spinning LiDAR feeding a W x H range image
        _SCAN_CACHE[key] = synth.generate_scan(k, n_azimuth=width, height=height, semantics=semantics)
    return _SCAN_CACHE[key]


@pytest.fixture(scope="session")
def scans():
    return get_scan


def bits(a):
    """bit pattern view of a float32 array (NaNs compare equal to themselves)"""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype == np.float32:
        ne = bits(a) != bits(b)
    else:
        ne = a != b
    n = int(np.count_nonzero(ne))
    assert n == 0, f"{what}: {n} of {ne.size} elements differ; first at {np.argwhere(ne)[:5].tolist()}"
