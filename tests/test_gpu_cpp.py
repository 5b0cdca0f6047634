"""GPU tests of the compiled-language bindings (run with -m gpu on an MI355X): the C++ adapter
(include/suma_adapter.hpp, what a maintainer compiles into src/core) and the C example (examples/odometry.c) run on
scans read from KITTI-style .bin files and must produce the same pose bits as the ctypes mirror / pipeline."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import params_with_size

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W = 4, 900


@pytest.fixture(scope="module")
def hip():
    from semantic_suma_amd import core
    core.lib()
    return core


@pytest.fixture(scope="module")
def scan_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("velodyne")
    for k in range(N):
        pts = get_scan(k, W, False)[0].copy()
        pts[:, 3] = 1.0
        pts.astype("<f4").tofile(str(d / f"{k:06d}.bin"))
    return str(d)


def mul4(A, B):
    """4x4 product in the operation order of the driver's mul4 (numpy's matmul may fuse / reorder)"""
    C = np.zeros((4, 4))
    for r in range(4):
        for c in range(4):
            C[r, c] = ((A[r, 0] * B[0, c] + A[r, 1] * B[1, c]) + A[r, 2] * B[2, c]) + A[r, 3] * B[3, c]
    return C


def build(hip, tmp, src, lang):
    exe = os.path.join(tmp, os.path.basename(src).split(".")[0])
    libdir = os.path.dirname(hip.LIB_PATH)
    cc = ["g++", "-std=c++11", "-O1"] if lang == "c++" else ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O1"]
    subprocess.check_call(cc + ["-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-lsuma_hip",
                                "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_adapter_runs_the_reference_call_sequence(hip, scan_dir, tmp_path):
    exe = build(hip, str(tmp_path), os.path.join(ROOT, "tests", "cpp", "adapter_driver.cpp"), "c++")
    out = subprocess.check_output([exe, scan_dir, str(N), str(W)], timeout=120).decode().strip().splitlines()
    assert out[-1].startswith("statistics ok"), out[-1]  # SurfelMapping::Stats keys of suma_hip::SurfelMapping
    out = out[:-1]
    assert len(out) == N
    # the same sequence through the ctypes mirror classes
    p = params_with_size(W, max_iterations=10, label_offset=0, prob_offset=0)
    pf = params_with_size(W, max_iterations=10, label_offset=0, prob_offset=0, icp_max_distance=p.fallback_max_distance,
                          icp_max_angle=p.fallback_max_angle)
    ctx = hip.Context(p)
    pre, smap, gn = hip.Preprocessing(ctx), hip.SurfelMap(ctx), hip.LieGaussNewton(ctx)
    objective, recovery = hip.Frame2Model(ctx), hip.Frame2Model(ctx, pf)
    current, last, model = hip.Frame(ctx, W, 64), hip.Frame(ctx, W, 64), hip.Frame(ctx, W, 64)
    pose, increment = np.eye(4), np.eye(4)
    for k in range(N):
        pts = np.fromfile(os.path.join(scan_dir, f"{k:06d}.bin"), dtype="<f4").reshape(-1, 4)
        current, last = last, current
        pre.process(pts, current, None, None, k)
        smap.render(pose, pose, model, -2.0)
        fb_outlier = 0
        if k > 0:
            objective.setData(current, smap.newMapFrame())
            gn.minimize(objective, increment)
            inc = gn.pose().copy()
            recovery.setData(current, last)
            gn.minimize(recovery, increment)
            fb_outlier = recovery.outlier()
            objective.initialize(inc)
            objective.jacobianProducts()
            increment = inc
            pose = mul4(pose, increment)
        smap.update(pose, current)
        cols = out[k].split()
        got = np.array([struct.unpack("<d", bytes.fromhex(h)[::-1])[0] for h in cols[1:17]]).reshape(4, 4).T
        assert np.array_equal(got, pose), f"scan {k}: pose bits differ between the C++ adapter and the ctypes mirror"
        _, first, n_data = smap.getDataSurfels()
        assert [int(v) for v in cols[17:22]] == [smap.size(), objective.valid(), objective.outlier(), fb_outlier, n_data]
        assert " ".join(cols[22:]) == "no error"
    assert pose[0, 3] > 2.0  # it tracked: ~1.1 m per scan


def test_c_example_runs_on_kitti_style_files(hip, scan_dir, tmp_path):
    exe = build(hip, str(tmp_path), os.path.join(ROOT, "examples", "odometry.c"), "c")
    out = subprocess.check_output([exe, scan_dir, str(N)], timeout=120, stderr=subprocess.DEVNULL).decode().strip().splitlines()
    assert len(out) == N
    p = params_with_size(2048)  # the example uses default.xml with 2048-column images
    pipe = hip.SurfelMapping(p)
    for k in range(N):
        pts = np.fromfile(os.path.join(scan_dir, f"{k:06d}.bin"), dtype="<f4").reshape(-1, 4)
        z = np.zeros(pts.shape[0], np.float32)
        pipe.processScan(pts, z, z, fixed_iterations=0)
        want = pipe.getCurrentPose()[:3, :].reshape(-1)
        got = np.array([float(v) for v in out[k].split()])
        assert np.allclose(got, want, rtol=0, atol=1e-8 * max(1.0, np.abs(want).max())), f"scan {k}"  # %.9g print


def test_gl_interop_recipe_on_the_gpu_box(hip, tmp_path):
    """the same program with a HIP device present and still no GL context: the registration is refused, nothing crashes"""
    exe = os.path.join(str(tmp_path), "gl_interop")
    libdir = os.path.dirname(hip.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "gl_interop.cpp"), "-o", exe, "-L",
                           libdir, "-lsuma_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "without a GL context: hipError" in out.stdout, out.stdout + out.stderr
