"""CPU tests of the phase API of the scan pipeline on the oracle side (TEST INFRASTRUCTURE), and of the host math the
loop-closure hooks add.  The GPU legs (HIP pipeline == oracle, bit for bit) are in tests/test_gpu_phases.py."""
import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import params_with_size

import loop_scenario as ls

W, H = 360, 32


def test_phases_equal_the_single_call(oracle_lib):
    """begin_scan + update_pose + update_map IS process_scan (SurfelMapping.cpp:175-204 without the hooks)"""
    p = params_with_size(W, H)
    a, b = oracle_lib.OraclePipeline(p), oracle_lib.OraclePipeline(p)
    for k in range(4):
        pts, lab, prob, _ = get_scan(k, W, True, H)
        a.process_scan(pts, lab, prob, fixed_iterations=6)
        b.begin_scan(pts, lab, prob)
        b.update_pose(6)
        b.update_map()
        assert np.array_equal(a.pose(), b.pose())
        assert a.last_stats().as_dict() == b.last_stats().as_dict()
        assert a.ctx.map_surfels().tobytes() == b.ctx.map_surfels().tobytes()
    # lastPose_old_ (:456) is the previous scan's currentPose_old_
    assert np.array_equal(b.get_pose(1), b.pose()) and np.array_equal(b.get_pose(2), b.pose())
    assert np.array_equal(b.get_pose(3), b.get_pose(4))


def test_se3_log_is_the_reference_function(oracle_lib):
    """ora_se3_log against SE3::log of the reference's lie_algebra.cpp compiled where it lies (oracle/_ref)"""
    pyref = pytest.importorskip("oracle.pyref")
    if not pyref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    rng = np.random.default_rng(5)
    for k in range(200):
        x = rng.uniform(-1, 1, 6) * (1e-6 if k % 10 == 0 else 1.0)
        x[3:] *= 0.5 if k % 3 else 3.0
        T = oracle_lib.se3_exp(x)
        a, b = oracle_lib.se3_log(T), pyref.se3_log(T)
        assert np.allclose(a, b, rtol=0, atol=1e-13), (k, a - b)
    assert np.array_equal(oracle_lib.se3_log(np.eye(4)), np.zeros(6))


def test_scripted_loop_closing_run(oracle_lib):
    """the script of tests/loop_scenario.py on the oracle alone: the closure is found one lap later, tracked, the
    optimised trajectory is integrated -- and each hook really changes what the pipeline computes"""
    p = params_with_size(W, H)
    lap = ls.lap_scans()
    k_detect = lap + 15
    n = k_detect + 11
    with_hooks = oracle_lib.OraclePipeline(p, threads=4)
    log = ls.run([ls.OraclePipe(with_hooks)], W, H, n, k_detect, 4, k_detect + 8, iterations=6)
    assert log["verify"] is not None and any(g["passed"] for g in log["verify"]), "no initial guess passed the gates"
    g = next(g for g in log["verify"] if g["passed"])
    assert g["composed"]["valid"] > 1000 and np.all(np.isfinite(g["JtJ"]))
    assert len(log["tracks"]) == 4 and any(t["passed"] for t in log["tracks"])
    for t in log["tracks"]:
        assert 0.0 <= t["increment_difference"] < 1.0
    assert log["integrated"] and log["moved_pose_old"] >= 2
    # the same scans without hooks end somewhere else: pose_old feeds the next render, integrate moves the pose
    plain = oracle_lib.OraclePipeline(p, threads=4)
    for k in range(n):
        plain.process_scan(*ls.scan(k, W, H), fixed_iterations=6)
    assert not np.array_equal(plain.pose(), with_hooks.pose())
    assert plain.ctx.map_surfels().tobytes() != with_hooks.ctx.map_surfels().tobytes()
    # odometry stayed on the circle
    gt = np.linalg.inv(ls.circle_pose(0)) @ ls.circle_pose(n - 1)
    assert np.linalg.norm((np.linalg.inv(with_hooks.pose()) @ gt)[:3, 3]) < 1.0
