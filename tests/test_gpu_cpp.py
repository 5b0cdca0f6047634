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


def test_bench_reads_a_kitti_format_directory(hip, tmp_path):
    """SUMA_KITTI_DIR (KITTIReader.cpp:136-203): bench.py's KITTI branch on a `sequences/00` directory in
    the dataset's own formats -- velodyne/*.bin (N x 4 float32: x, y, z, remission), labels/*.label (uint32, class id in
    the lower 16 bits), calib.txt.  No KITTI scan exists on either machine, so the files are the synthetic sequence
    written in those formats: what is exercised is the reader path end to end (the remission column dropped, labels
    remapped as RangeNet++ reports them, label / prob offsets of quirk B-1 switched off), the line says data = kitti,
    and the same files through the Python mirror give the pose the in-memory scans give."""
    import json
    import sys
    from semantic_suma_amd import kitti
    W, N = 900, 30
    root = tmp_path / "sequences" / "00"
    (root / "velodyne").mkdir(parents=True)
    (root / "labels").mkdir()
    mem = []
    for k in range(N):
        pts, lab, prob, _ = get_scan(k, W, True)
        raw = pts.copy()
        raw[:, 3] = 0.37  # remission: the reader must drop it
        raw.astype("<f4").tofile(str(root / "velodyne" / f"{k:06d}.bin"))
        (lab.astype("<u4") | np.uint32(7 << 16)).astype("<u4").tofile(str(root / "labels" / f"{k:06d}.label"))
        mem.append((pts, lab))
    (root / "calib.txt").write_text("Tr: 0 -1 0 0 0 0 -1 0 1 0 0 0\n")
    env = dict(os.environ, SUMA_KITTI_DIR=str(root))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", str(W), "--steps", "5", "--warmup", "2",
                          "--preroll", "12", "--cpu-scans", "0", "--adapter-scans", "0", "--no-kernel-events",
                          "--host-vector-scans", "5", "--no-loop-closure", "--no-reference-mode"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["data"] == "kitti" and d["value"] > 100 and d["config"]["points_per_scan"] > 10000
    # the same files through the reader and the Python mirror against the in-memory scans they were written from
    seq = kitti.Sequence(str(root))
    assert len(seq) == N
    p = params_with_size(W, label_offset=0, prob_offset=0)
    a, b = hip.SurfelMapping(p), hip.SurfelMapping(p)
    for k in range(6):
        pts, lab, prob = seq[k]
        assert np.array_equal(pts, mem[k][0]) and np.array_equal(lab, mem[k][1]) and (prob == 1).all()
        a.processScan(pts, lab, prob, fixed_iterations=6)
        b.processScan(mem[k][0], mem[k][1], np.ones_like(prob), fixed_iterations=6)
    assert np.array_equal(a.getCurrentPose(), b.getCurrentPose()) and a.getCurrentPose()[0, 3] > 3.0
