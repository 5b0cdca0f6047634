"""End to end: SurfelMapping::processScan with the reference's own shaders in a REAL OpenGL (oracle/glpipeline.py: Mesa
llvmpipe executes every pass K1-K12 of /root/reference/src/shader, numpy restates the host code between the passes)
against the CPU oracle -- the checker the HIP path is bit-identical to -- on identical scans.

This is the task's acceptance criterion -- "pose delta within 1e-4 m / 1e-5 rad per ICP iteration" of the reference's
OpenGL path -- measured on an actual GL implementation.  CPU only; skipped where Mesa or /root/reference is absent."""
import math
import os

import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import params_with_size

W, H, ITER = 900, 64, 10


def pose_delta(A, B):
    D = np.linalg.inv(np.asarray(A, dtype=np.float64)) @ np.asarray(B, dtype=np.float64)
    return float(np.linalg.norm(D[:3, 3])), float(math.acos(max(-1.0, min(1.0, 0.5 * (np.trace(D[:3, :3]) - 1.0)))))


@pytest.fixture(scope="module")
def glp():
    from oracle import glref, pyref
    if not glref.available():
        pytest.skip("no software GL (Mesa swrast_dri.so + DRI headers) on this machine")
    if not pyref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle import glpipeline
    return glpipeline


@pytest.mark.parametrize("which", ["driver", "detmath"])
def test_per_iteration_pose_increments_of_the_gl_path(glp, oracle_lib, which):
    """Teacher-forced, per ICP iteration, in two variants (round-5 advisor: both are kept side by side).

    "driver" -- the reference GL path AS SHIPPED: the reference's shader text and the GL implementation's own asin / acos /
    atan (Mesa llvmpipe here; no vendor GL exists on either machine).  This is the check that does not depend on anything
    this repository specifies.  Asserted at the bounds it has carried since round 4: every step within 2e-3 m / 3e-4 rad of
    the oracle's, the median within 3e-4 m / 5e-5 rad.  north_star's 1e-4 m / 1e-5 rad is NOT met against it (32 of 40
    steps are beyond 1e-4 m): llvmpipe's asin is up to 3.9e-4 rad off -- GLSL leaves the accuracy of the angle functions
    to the implementation -- and moves ~10 pairs per iteration across a gate (tests/test_gl_controls.py).

    "detmath" -- a CONTROL, not the reference as shipped: the same shader text with its three angle functions #defined to
    the specified ones (oracle/glref.py::DETMATH_PRELUDE, checked bit for bit against include/suma_detmath.h), i.e. both
    sides share their transcendentals and what remains is everything else a real GL does (rasterisation, fp32 blending
    in draw order, texture filtering).  Here the acceptance tolerance holds with a margin of 70: the claim is "within
    1e-4 m / 1e-5 rad per iteration of the reference's shader text UNDER SPECIFIED TRANSCENDENTALS".

    Per step: from the oracle's pose before iteration k, ONE Gauss-Newton
    step with the reference's Frame2Model_jacobians shaders executed by llvmpipe, on the oracle's frames, against the
    oracle's pose after iteration k -- the pose the HIP path reproduces bit for bit.  north_star: "pose delta within
    1e-4 m / 1e-5 rad per ICP iteration" of the reference OpenGL path.

    Round 4 ran this with the DRIVER'S asin / atan (llvmpipe: up to 3.9e-4 rad off, a twentieth of an image row) and saw
    0.4 .. 10 x 10^-4 m; round 5's controls (tests/test_gl_controls.py) show that this was the driver's asin alone moving
    ~10 pairs per iteration across a gate -- GL against itself under a permuted draw order differs by 4e-7 m, i.e. fp32
    blend order is three orders of magnitude smaller.  Here the reference's shader text runs unchanged with its three angle
    functions #defined to the specified ones (oracle/glref.py::DETMATH_PRELUDE -- GLSL leaves their accuracy to the
    implementation): every one of the 40 steps is asserted within 1e-4 m / 1e-5 rad (measured: 3e-7 m / 3e-8 rad)."""
    p = params_with_size(W)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    with glp.gl.transcendentals(which):
        k6 = glp.gl.Jacobians(p)
    steps = []
    for k in range(5):
        pts, lab, prob, _ = get_scan(k, W, True)
        if k >= 1:
            ora = op.ctx
            cur = ora.preprocess(pts, lab, prob, k, ora.frame())
            pose32 = op.pose().astype(np.float32)
            out = ora.frame(model=True)
            # the model frame the minimisation of scan k will see (SurfelMapping.cpp:344-351, 384)
            ct = p.confidence_threshold if k >= 10 else float(np.float32((1.0 - k / 10.0) * math.log(0.1 / 0.9) + np.float32(k / 10.0) * np.float32(p.confidence_threshold)))
            ora.map_render(pose32, pose32, ct, out)
            model = ora.map_frame(1)
            pp = params_with_size(W, max_iterations=ITER, stopping_threshold=0.0, delta=0.0)
            ora.set_params(pp)
            _, hist, st = ora.minimize(cur, model, op.last_increment(), history_cap=ITER + 1)
            ora.set_params(p)
            assert hist.shape[0] == ITER + 1
            cm, mm = [cur.map(m) for m in range(3)], [model.map(m) for m in range(3)]
            for it in range(ITER):
                b = k6.run(cm, mm, hist[it], it)
                dx = np.linalg.solve(b[:36].reshape(6, 6).astype(np.float64), -b[36:42].astype(np.float64))
                Tn = glp.pyref.se3_exp(dx) @ hist[it]
                steps.append((k, it) + pose_delta(Tn, hist[it + 1]) + (st.valid,))
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)
    for row in steps:
        print("scan %d it %d: %.2e m %.2e rad (%d pairs)" % row)
    dts, drs = np.array([r[2] for r in steps]), np.array([r[3] for r in steps])
    assert len(steps) == 40
    print(f"per-iteration GL ({which} transcendentals) vs oracle, worst of {len(steps)} steps: {dts.max():.2e} m / "
          f"{drs.max():.2e} rad, median {np.median(dts):.2e} m / {np.median(drs):.2e} rad, "
          f"{int((dts > 1e-4).sum())} steps beyond 1e-4 m")
    if which == "detmath":
        assert dts.max() <= 1e-4 and drs.max() <= 1e-5, (dts.max(), drs.max())
    else:
        assert dts.max() <= 2e-3 and drs.max() <= 3e-4, (dts.max(), drs.max())
        assert np.median(dts) <= 3e-4 and np.median(drs) <= 5e-5, (np.median(dts), np.median(drs))


def test_free_running_gl_pipeline_against_the_oracle(glp, oracle_lib):
    """Free running: eight scans (8.8 m) through the GL path -- preprocessing, rendering, ten Gauss-Newton iterations,
    index map, radius map, surfel update with transform feedback, new surfels, active-area copy: all the reference's
    shaders, all in llvmpipe -- and through the oracle.  Nothing is teacher-forced: each side's maps and poses feed its
    own next scan.  Measured: the trajectories part by 2 cm in the cold minimisation of scan 1 (start = identity, 1.1 m
    of motion: from identical inputs the GL path alone ends 5 mm from the oracle there, its per-step noise amplified by
    the gates) and stay 1 - 2 cm / 2 - 5 x 10^-4 rad apart afterwards, increments agree to 3 - 8 mm, the maps to 0.1 % in
    size.  Asserted: 5 cm / 2e-3 rad, map sizes within 0.3 %."""
    p = params_with_size(W)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    g = glp.GLPipeline(p)
    log = []
    for k in range(8):
        pts, lab, prob, _ = get_scan(k, W, True)
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)
        g.process_scan(pts, lab, prob, ITER)
        dt, dr = pose_delta(op.pose(), g.current_pose)
        it_, ir_ = pose_delta(op.last_increment(), g.last_increment)
        print(f"  increment of scan {k}: {it_:.2e} m / {ir_:.2e} rad apart")
        su, sn = op.ctx.map_counts()
        n_o, n_g = op.ctx.map_size(), g.counts["map"]
        log.append((k, dt, dr, n_o, n_g, su, g.counts["updated"], sn, g.counts["new"]))
        assert dt <= 5e-2 and dr <= 2e-3, f"scan {k}: {dt:.2e} m / {dr:.2e} rad apart"
        assert abs(n_o - n_g) <= 0.003 * n_o + 5, f"scan {k}: map {n_g} surfels in GL, {n_o} in the oracle"
    for row in log:
        print("scan %d: %.2e m %.2e rad | map %d / %d | updated %d / %d | new %d / %d" % row)
    assert np.linalg.norm(g.current_pose[:3, 3]) > 5.0, "the sensor must have moved"


def test_free_running_gl_pipeline_at_the_bench_geometry(glp, oracle_lib):
    """The same free-running comparison at 64 x 2048, the geometry bench.py times (BASELINE configs[1]): six scans.  With
    2.3 times the texels the fp32 blend-order noise of the GL path averages out further -- measured: the trajectories stay
    within 1.6 mm / 2.3e-4 rad of each other (9 x 10^-4 m after the cold second scan), the maps within 0.1 % in size
    (156 393 / 156 310 surfels).  Asserted: 5 mm / 5e-4 rad, 0.3 %."""
    Wb = 2048
    p = params_with_size(Wb)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    g = glp.GLPipeline(p)
    for k in range(6):
        pts, lab, prob, _ = get_scan(k, Wb, True)
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)
        g.process_scan(pts, lab, prob, ITER)
        dt, dr = pose_delta(op.pose(), g.current_pose)
        n_o, n_g = op.ctx.map_size(), g.counts["map"]
        print("scan %d: %.2e m %.2e rad | map %d / %d" % (k, dt, dr, n_o, n_g))
        assert dt <= 5e-3 and dr <= 5e-4, f"scan {k}: {dt:.2e} m / {dr:.2e} rad apart"
        assert abs(n_o - n_g) <= 0.003 * n_o + 5
    assert np.linalg.norm(g.current_pose[:3, 3]) > 4.0


def test_gl_pipeline_through_the_submap_window(glp, oracle_lib):
    """The GL path with the submap window moving: 1.9 m half-width tiles in a 17 x 17 window, 1.1 m per scan, so that every other scan a
    row of tiles leaves the window, is parked tile by tile through extract_surfels.vert + transform feedback, and the copy
    extent grows while tiles wait (SurfelMap.cpp:667-677, 708-824).  Origins and the extraction order are the oracle's at
    every scan (4 shifts, 14 tiles parked in 16 scans), the parked tiles hold the same number of surfels to 1 %, the maps
    agree to 0.1 % in size, the trajectories stay within 5 cm / 3e-3 rad (measured 3.5 cm / 1.6e-3 rad)."""
    p = params_with_size(W, submap_extent=1.9, submap_dimension=8)  # 1.1 x 1.9 m: no step of the path lands near a shift threshold
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    g = glp.GLPipeline(p)
    parked = 0
    for n, k in enumerate(range(16)):
        pts, lab, prob, _ = get_scan(k, W, True)
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)
        g.process_scan(pts, lab, prob, ITER)
        dt, dr = pose_delta(op.pose(), g.current_pose)
        ora = op.ctx
        assert tuple(g.origin) == tuple(ora.map_submap_origin()), f"scan {n}: submap origin"
        count, (ei, ej) = ora.map_last_extraction()
        assert count == g.extractions and len(g.extraction) == ora.map_pending_extractions(), f"scan {n}: extraction list"
        if count:
            assert (int(ei), int(ej)) == g.last_extraction
            want, got = ora.map_cache_tile(int(ei), int(ej)).shape[0], g.cache[g.last_extraction].shape[0]
            assert abs(want - got) <= 0.01 * want + 10, f"scan {n}: tile ({ei},{ej}) {got} surfels in GL, {want} in the oracle"
            parked = max(parked, want)
        n_o, n_g = ora.map_size(), g.surfels.shape[0]
        print("step %d (scan %d): %.2e m %.2e rad | origin %s | map %d / %d | extractions %d" % (n, k, dt, dr, tuple(g.origin), n_o, n_g, count))
        assert dt <= 5e-2 and dr <= 3e-3, f"scan {n}: {dt:.2e} m / {dr:.2e} rad apart"
        assert abs(n_o - n_g) <= 0.005 * n_o + 10
    assert g.extractions >= 8 and tuple(g.origin) != (0, 0) and parked > 20, parked


@pytest.mark.parametrize("which,tol_m", [("driver", 2e-1), ("detmath", 1e-1)])
def test_gl_pipeline_takes_the_fallback_like_the_oracle(glp, oracle_lib, which, tol_m):
    """A jump in the motion (two scans skipped) trips the frame-to-frame fallback minimisation (SurfelMapping.cpp:434-449)
    in the GL path at the same scan as in the oracle, and the two trajectories stay together through it -- with the
    driver's own angle functions (a cold minimisation from 3.3 m away amplifies its asin: 11 cm after the jump) and with
    the specified ones (oracle/glref.py::DETMATH_PRELUDE: 5 cm; what is left is the model render's attribute interpolation)."""
    p = params_with_size(W)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    with glp.gl.transcendentals(which):
        g = glp.GLPipeline(p)
    for n, k in enumerate([0, 1, 2, 3, 6, 7]):
        pts, lab, prob, _ = get_scan(k, W, True)
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)
        g.process_scan(pts, lab, prob, ITER)
        dt, dr = pose_delta(op.pose(), g.current_pose)
        print("step %d (scan %d): %.2e m %.2e rad | track loss %d / %d" % (n, k, dt, dr, op.track_loss(), g.track_loss))
        assert op.track_loss() == g.track_loss, f"step {n}: fallback decision"
        assert dt <= tol_m and dr <= 5e-3, f"step {n}: {dt:.2e} m / {dr:.2e} rad apart"
    assert g.track_loss >= 1, "the sequence was meant to trip the fallback"


@pytest.mark.parametrize("fixture,tol_m,tol_rad", [("gl_pipeline_900x64.npz", 5e-2, 2e-3), ("gl_pipeline_2048x64.npz", 5e-3, 5e-4)])
def test_oracle_agrees_with_the_committed_gl_trajectories(oracle_lib, fixture, tol_m, tol_rad):
    """The committed GL-made trajectories (tests/golden/make_gl_pipeline_golden.py; no Mesa needed here) against the
    oracle on the same seeded scans -- the CPU twin of tests/test_gpu_gl_golden.py."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    Wf, n, iters = int(z["W"]), int(z["scans"]), int(z["iterations"])
    op = oracle_lib.OraclePipeline(params_with_size(Wf, int(z["H"])), threads=max(1, min(8, os.cpu_count() or 1)))
    for k in range(n):
        pts, lab, prob, _ = get_scan(k, Wf, True)
        op.process_scan(pts, lab, prob, fixed_iterations=iters)
        dt, dr = pose_delta(z["poses"][k], op.pose())
        assert dt <= tol_m and dr <= tol_rad, f"scan {k}: {dt:.2e} m / {dr:.2e} rad from the GL path"
        gl_map = int(z["counts"][k][0])
        assert abs(op.ctx.map_size() - gl_map) <= 0.003 * gl_map + 5


def test_oracle_agrees_with_the_committed_gl_gn_steps(oracle_lib):
    """CPU twin of tests/test_gpu_gl_golden.py::test_acceptance_line_hip_against_the_reference_gl_path_per_iteration: the
    committed GL-made Gauss-Newton steps (tests/golden/gl_gn_steps_900x64.npz; no Mesa needed here) against the oracle's
    teacher-forced minimisations -- same poses before every iteration bit for bit, the pose after within 1e-4 m /
    1e-5 rad of the reference's shaders in OpenGL."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gl_gn_steps_900x64.npz"))
    Wf, n, iters = int(z["W"]), int(z["scans"]), int(z["iterations"])
    p = params_with_size(Wf)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    row = 0
    for k in range(n):
        pts, lab, prob, _ = get_scan(k, Wf, True)
        if k >= 1:
            ora = op.ctx
            cur = ora.preprocess(pts, lab, prob, k, ora.frame())
            pose32 = op.pose().astype(np.float32)
            out = ora.frame(model=True)
            ct = float(np.float32((1.0 - k / 10.0) * math.log(0.1 / 0.9) + np.float32(k / 10.0) * np.float32(p.confidence_threshold)))
            ora.map_render(pose32, pose32, ct, out)
            ora.set_params(params_with_size(Wf, max_iterations=iters, stopping_threshold=0.0, delta=0.0))
            _, hist, _ = ora.minimize(cur, ora.map_frame(1), op.last_increment(), history_cap=iters + 1)
            ora.set_params(p)
            for it in range(iters):
                assert np.array_equal(hist[it], z["pose_before"][row])
                dt, dr = pose_delta(z["pose_after_gl"][row], hist[it + 1])
                assert dt <= 1e-4 and dr <= 1e-5, f"scan {k} iteration {it}: {dt:.2e} m / {dr:.2e} rad"
                row += 1
        op.process_scan(pts, lab, prob, fixed_iterations=iters)
    assert row == 40
