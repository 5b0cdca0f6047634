"""The deterministic math specification (include/suma_detmath.h) against libm: accuracy in ulp on the
domains the pipeline uses.  Bit equality between host and gfx950 is checked on the GPU by tools/fpcheck.hip
(and implicitly by every bit-exact GPU parity test)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("detmath") / "detmath_shim.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off",
                           os.path.join(HERE, "detmath_shim.c"), "-o", so, "-lm"])
    return C.CDLL(so)


def run1(shim, name, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    getattr(shim, name)(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size)
    return y


def ulp_err(y, ref):
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(y.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)


@pytest.mark.parametrize("name,fn,lo,hi,tol", [
    ("t_atan", np.arctan, -100.0, 100.0, 2.5), ("t_asin", np.arcsin, -1.0, 1.0, 3.0),
    ("t_acos", np.arccos, -1.0, 1.0, 3.0), ("t_sin", np.sin, -3.2, 3.2, 2.5), ("t_cos", np.cos, -3.2, 3.2, 2.5),
    ("t_exp", np.exp, -20.0, 5.0, 2.5), ("t_log", np.log, 1e-6, 100.0, 2.5), ("t_sqrt", np.sqrt, 0.0, 1e4, 0.5001),
])
def test_accuracy_vs_libm(shim, name, fn, lo, hi, tol):
    rng = np.random.default_rng(0)
    x = rng.uniform(lo, hi, 200000).astype(np.float32)
    y = run1(shim, name, x)
    err = ulp_err(y, fn(x.astype(np.float64)))
    # near zeros of sin/cos the absolute error is what matters (argument reduction); exclude |ref| tiny
    mask = np.abs(fn(x.astype(np.float64))) > 1e-3
    assert err[mask].max() <= tol, f"{name}: {err[mask].max()} ulp"


def test_atan2_quadrants_and_floor_round(shim):
    rng = np.random.default_rng(1)
    y = rng.uniform(-50, 50, 100000).astype(np.float32)
    x = rng.uniform(-50, 50, 100000).astype(np.float32)
    r = np.empty_like(x)
    shim.t_atan2(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), x.size)
    assert ulp_err(r, np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 4.0  # division + atan + quadrant add
    v = np.concatenate([rng.uniform(-3000, 3000, 100000), [0.0, -0.0, 0.5, -0.5, 2047.999, -1e-30, 1e9, -1e9]]).astype(np.float32)
    assert np.array_equal(run1(shim, "t_floor", v), np.floor(v))
    near = np.round(rng.uniform(0, 255, 1000)).astype(np.float32) + rng.uniform(-0.2, 0.2, 1000).astype(np.float32)
    assert np.array_equal(run1(shim, "t_round", near), np.round(near))
    # GLSL round() at .5: to even, as the GL implementations do (76.5 and 178.5 are pack()'s ties, color.glsl:34-36)
    ties = np.array([0.5, 1.5, 2.5, 76.5, 77.5, 178.5, 179.5, -0.5, -1.5, -2.5, 8388607.5, 0.49999997, 0.50000006], dtype=np.float32)
    assert np.array_equal(run1(shim, "t_round", ties), np.round(ties))
    assert run1(shim, "t_round", np.array([76.5, 178.5], np.float32)).tolist() == [76.0, 178.0]


def test_double_sincos(shim):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-10, 10, 100000), rng.uniform(-1e-3, 1e-3, 10000), [0.0, 1e-300]])
    s, c = np.empty_like(x), np.empty_like(x)
    shim.t_sin_d(x.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), x.size)
    shim.t_cos_d(x.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), x.size)
    assert np.abs(s - np.sin(x)).max() < 3e-16 and np.abs(c - np.cos(x)).max() < 3e-16
