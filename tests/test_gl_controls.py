"""The two CONTROLS behind the acceptance line "pose delta within 1e-4 m / 1e-5 rad per ICP iteration of the reference
OpenGL path" (round-4 review, item 1).  Round 4 measured one Gauss-Newton step of the reference's
Frame2Model_jacobians.{vert,geom,frag} in a real OpenGL (Mesa llvmpipe) 0.4 .. 10 x 10^-4 m / 0.3 .. 11 x 10^-5 rad from
the oracle's step and ATTRIBUTED that to fp32 blend order + the driver's inaccurate asin.  These tests separate the two:

  (a) GL AGAINST ITSELF: the very same draw with the point list reversed / shuffled.  Blending happens in primitive order,
      so this is the blend-order noise of the reference path, measured: <= 4e-7 m / 5e-8 rad per step.  It is NOT what
      round 4 saw.
  (b) GL WITH SPECIFIED TRANSCENDENTALS: the reference's shader text unchanged, but asin / acos / atan #defined to the
      Cephes kernels of include/suma_detmath.h restated in GLSL (oracle/glref.py::DETMATH_PRELUDE; GLSL leaves the
      accuracy of the angle functions to the implementation, llvmpipe's asin is off by up to 3.9e-4 rad).  One step of
      THAT GL path lies <= 3e-7 m / 3e-8 rad from the oracle's step on all 40 + 20 steps, with identical valid / outlier
      counts: the whole round-4 discrepancy was the driver's asin moving ~10 pairs per iteration across a gate, and the
      acceptance tolerance is met with a margin of 300.

With the same prelude in every pass the fixed-function half is pinned much harder than round 4's "98-99 % of the texels":
gen_vertexmap (K1) is bit-equal on ALL texels, gen_indexmap (K7) names the same surfel on all texels but one.

CPU only; skipped where Mesa or /root/reference is absent.  REPORT=path writes the measured table as JSON
(profiles/r05_gl_acceptance_controls.json was made that way)."""
import json
import math
import os

import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import params_with_size

H, ITER = 64, 10
REPORT = {}


def pose_delta(A, B):
    D = np.linalg.inv(np.asarray(A, dtype=np.float64)) @ np.asarray(B, dtype=np.float64)
    return float(np.linalg.norm(D[:3, 3])), float(math.acos(max(-1.0, min(1.0, 0.5 * (np.trace(D[:3, :3]) - 1.0)))))


@pytest.fixture(scope="module")
def gl():
    from oracle import glref, pyref
    if not glref.available():
        pytest.skip("no software GL (Mesa swrast_dri.so + DRI headers) on this machine")
    if not pyref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    yield glref
    if os.environ.get("REPORT") and REPORT:
        info = glref.limits()
        REPORT["gl"] = {"version": info["version"], "renderer": info["renderer"]}
        with open(os.environ["REPORT"], "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)


def teacher_forced_steps(oracle_lib, W, scans):
    """(scan, iteration, data maps, model maps, pose before, the oracle's pose after, its valid / outlier counts) for
    every Gauss-Newton iteration of scans 1 .. scans-1 of the oracle's own run -- the set-up of
    tests/test_gl_pipeline.py::test_per_iteration_pose_increments_of_the_gl_path"""
    p = params_with_size(W)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    for k in range(scans):
        pts, lab, prob, _ = get_scan(k, W, True)
        if k >= 1:
            ora = op.ctx
            cur = ora.preprocess(pts, lab, prob, k, ora.frame())
            pose32 = op.pose().astype(np.float32)
            out = ora.frame(model=True)
            ct = p.confidence_threshold if k >= 10 else float(np.float32((1.0 - k / 10.0) * math.log(0.1 / 0.9) + np.float32(k / 10.0) * np.float32(p.confidence_threshold)))
            ora.map_render(pose32, pose32, ct, out)
            model = ora.map_frame(1)
            ora.set_params(params_with_size(W, max_iterations=ITER, stopping_threshold=0.0, delta=0.0))
            _, hist, _ = ora.minimize(cur, model, op.last_increment(), history_cap=ITER + 1)
            ora.set_params(p)
            cm, mm = [cur.map(m).copy() for m in range(3)], [model.map(m).copy() for m in range(3)]
            for it in range(ITER):
                _, _, _, _, st = ora.jacobian_products(cur, model, hist[it], it)
                yield k, it, cm, mm, hist[it], hist[it + 1], (int(st.valid), int(st.outlier))
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)


def gn_step(pyref, b, T):
    """LieGaussNewton::step on the 48 floats Frame2Model.cpp:214-227 unpacks: JtJ.ldlt().solve(-Jtf), exp(delta) * pose"""
    dx = np.linalg.solve(b[:36].reshape(6, 6).astype(np.float64), -b[36:42].astype(np.float64))
    return pyref.se3_exp(dx) @ T


@pytest.mark.parametrize("W,scans", [(900, 5), (2048, 3)])
def test_gn_step_of_the_gl_path_with_and_without_the_drivers_asin(gl, oracle_lib, W, scans):
    """Control (b) and, beside it, the uncontrolled measurement of round 4 on the same steps.  Asserted on the
    specified-transcendentals variant: EVERY step within 1e-4 m / 1e-5 rad of the oracle's (the acceptance line; measured
    3e-7 m / 3e-8 rad, asserted at 2e-6 / 2e-7) and the same valid / outlier counts on every step.  The stock driver is
    only bounded loosely (2e-3 m / 3e-4 rad): it is llvmpipe's asin, not the reference, that is measured there."""
    from oracle import pyref
    p = params_with_size(W)
    stock = gl.Jacobians(p)
    with gl.transcendentals("detmath"):
        spec = gl.Jacobians(p)
    ref = pyref.Ref(p)
    rows = []
    for k, it, cm, mm, T, T_next, (valid, outlier) in teacher_forced_steps(oracle_lib, W, scans):
        b_stock, b_spec = stock.run(cm, mm, T, it), spec.run(cm, mm, T, it)
        blend, _ = ref.jacobians(cm, mm, T, it)  # the same shader text compiled by g++, fp32 blending in draw order
        d_stock, d_spec, d_blend = (pose_delta(gn_step(pyref, b, T), T_next) for b in (b_stock, b_spec, blend))
        rows.append(dict(scan=k, it=it, stock=d_stock, specified=d_spec, compiled_text_fp32_blend=d_blend, pairs=valid,
                         flips_stock=[int(b_stock[42]) - valid, int(b_stock[44]) - outlier],
                         flips_specified=[int(b_spec[42]) - valid, int(b_spec[44]) - outlier]))
        print("scan %d it %d | stock %.2e m %.2e rad (valid %+d outlier %+d) | specified %.2e m %.2e rad (valid %+d outlier %+d) | "
              "compiled text, fp32 blend %.1e m %.1e rad" % (k, it, *d_stock, *rows[-1]["flips_stock"], *d_spec,
                                                             *rows[-1]["flips_specified"], *d_blend))
    assert len(rows) == (scans - 1) * ITER
    worst = {key: (max(r[key][0] for r in rows), max(r[key][1] for r in rows)) for key in ("stock", "specified", "compiled_text_fp32_blend")}
    REPORT[f"gn_step_64x{W}"] = dict(steps=len(rows), worst_m_rad=worst, rows=rows,
                                     stock_steps_over_1e_4_m=sum(r["stock"][0] > 1e-4 for r in rows),
                                     stock_steps_over_1e_5_rad=sum(r["stock"][1] > 1e-5 for r in rows))
    print("worst of %d steps:" % len(rows), worst)
    # the acceptance line, on the reference's GLSL in a real GL with the specified angle functions
    assert worst["specified"][0] <= 1e-4 and worst["specified"][1] <= 1e-5
    assert worst["specified"][0] <= 2e-6 and worst["specified"][1] <= 2e-7, worst["specified"]
    assert all(r["flips_specified"] == [0, 0] for r in rows), "a pair changed sides of a gate"
    assert worst["stock"][0] <= 2e-3 and worst["stock"][1] <= 3e-4
    assert any(r["flips_stock"] != [0, 0] for r in rows)  # what the driver's asin does


def test_gl_against_itself_under_a_permuted_draw_order(gl, oracle_lib):
    """Control (a): blend-order noise of the reference path, measured on the reference path.  The same
    Frame2Model_jacobians draw with the 960 points in vbo order, reversed, and in three random orders (blending is applied
    in primitive order, GL 4.5 section 17.3.8, so the order of the list IS the order of the fp32 additions); stock
    driver transcendentals, so nothing but the order differs.  Measured: one GN step moves by <= 4e-7 m / 5e-8 rad
    (asserted 5e-6 / 5e-7), the counters (sums of 1.0) are identical -- three orders of magnitude below what round 4 put
    down to "fp32 blend order"."""
    from oracle import pyref
    W = 900
    p = params_with_size(W)
    base = gl.Jacobians(p)
    n = base.n
    rng = np.random.default_rng(7)
    others = [gl.Jacobians(p, order=np.arange(n)[::-1])] + [gl.Jacobians(p, order=rng.permutation(n)) for _ in range(3)]
    worst, rel = (0.0, 0.0), 0.0
    for k, it, cm, mm, T, _, _ in teacher_forced_steps(oracle_lib, W, 3):
        b0 = base.run(cm, mm, T, it)
        assert np.array_equal(b0, base.run(cm, mm, T, it)), "the same draw twice must give the same bits"
        T0 = gn_step(pyref, b0, T)
        for o in others:
            b = o.run(cm, mm, T, it)
            assert b[42] == b0[42] and b[44] == b0[44] and b[46] == b0[46]
            dt, dr = pose_delta(gn_step(pyref, b, T), T0)
            worst = (max(worst[0], dt), max(worst[1], dr))
            rel = max(rel, float(np.abs(b[:42] - b0[:42]).max() / np.abs(b0[:36]).max()))
    print("GL vs GL under permuted draw order: worst step difference %.2e m / %.2e rad, sums differ by %.1e of the largest entry" % (*worst, rel))
    REPORT["gl_vs_gl_permuted_order_64x900"] = dict(orders=1 + len(others), steps=2 * ITER, worst_m_rad=worst, sums_rel=rel)
    assert worst[0] <= 5e-6 and worst[1] <= 5e-7
    assert rel > 0.0, "the order must matter at all, or the control is void"


def test_the_map_passes_in_gl_with_specified_transcendentals(gl, oracle_lib):
    """The reference's gen_vertexmap (K1), gen_indexmap (K7), render_surfels (K4), update_surfels with real transform
    feedback (K9) and gen_surfels (K10) in llvmpipe, once with the driver's angle functions (the measurements of
    tests/test_gl_reference.py) and once with the specified ones, against the oracle / the compiled shader text on
    identical inputs.  With the driver's asin out of the way what a real GL and the oracle still disagree on is the
    fixed-function freedom alone (attribute interpolation of K4's quads, the last ulp of exp / log / normalize):
      K1  98.84 % of the texels bit-equal  ->  ALL 57 600
      K7  99.31 % name the same surfel     ->  all but ONE texel (37 054 / 37 055 occupied)
      K4  98.5 % carry the same surfel     ->  99.55 %
      K9  1 surfel survives on one side, 76 mask texels differ  ->  the same 101 184 survivors, ONE mask texel differs
      K10 the same 7 058 new surfels in the same order (both).
    i.e. "bit-exact surfel indices / counts" holds against the reference's shaders executed by a real OpenGL."""
    from oracle import pyref
    W = 900
    p = params_with_size(W)
    res = {}
    # K1
    ora = oracle_lib.Oracle(p)
    pts, lab, prob, _ = get_scan(3, W, True)
    fr = ora.preprocess(pts, lab, prob, 20, ora.frame())  # keep the frame alive: map() is a view of its buffer
    ov = fr.map(0).copy()
    for which in ("driver", "detmath"):
        with gl.transcendentals(which):
            gv, _ = gl.VertexMap(p).run(pts, lab, prob, 20)
        res[f"K1_{which}"] = int(np.sum(~np.all(ov.view(np.uint32) == gv.view(np.uint32), axis=-1)))
    # a 12-scan map, then scan 12 on top of it
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    for k in range(12):
        op.process_scan(*get_scan(k, W, True)[:3], fixed_iterations=ITER)
    ctx = op.ctx
    before, ts = ctx.map_surfels().copy(), ctx.map_timestamp()
    pose_r = op.pose().astype(np.float32).astype(np.float64)
    out = ctx.frame(model=True)
    ctx.map_render(pose_r, pose_r, 0.0, out)
    o4 = ctx.map_frame(1).map(0).copy()
    for which in ("driver", "detmath"):
        with gl.transcendentals(which):
            g4 = gl.SurfelRenderer(p).render(before, ctx.map_poses(ts + 1).reshape(-1, 16), pose_r, 0.0, ts - 100, False)[0]
        va, vb = o4[..., 3] > 0.5, g4[..., 3] > 0.5
        same = va & vb & np.all(np.abs(o4 - g4) <= 1e-4 * (1.0 + np.abs(o4)), axis=-1)
        res[f"K4_{which}"] = float(same.sum() / max(va.sum(), vb.sum()))
    op.process_scan(*get_scan(12, W, True)[:3], fixed_iterations=ITER)
    want7 = ctx.map_index_map()
    pose = op.pose().astype(np.float32)
    inv = np.eye(4, dtype=np.float32)
    inv[:3, :3], inv[:3, 3] = pose[:3, :3].T, -(pose[:3, :3].T @ pose[:3, 3])
    cur = op.frame(0)
    frame = (cur.vertex.copy(), cur.normal.copy(), cur.semantic.copy())
    poses = ctx.map_poses(ts + 2)
    pose_t = poses[ts].reshape(4, 4).T
    o_idx, o_rc = want7.astype(np.float32), ctx.map_radius_conf()
    ref = pyref.Ref(p)
    want9, want_mask, want_src = ref.update(before, poses, pose_t, ts, frame, o_rc, o_idx, sources=True)
    n = before.shape[0]
    mask4 = np.zeros((H, W, 4), np.float32)
    mask4[:, :, 0] = ctx.map_integrated() != 0
    want10 = ref.generate(frame, o_rc, mask4, pose_t, ts).view(np.float32).reshape(-1, 16)
    for which in ("driver", "detmath"):
        with gl.transcendentals(which):
            K7, K9, K10 = gl.IndexMap(p), gl.SurfelUpdate(p, tag_sources=True), gl.SurfelGenerate(p)
        got7, _ = K7.run(before, poses[: ts + 1].reshape(-1, 16), pose, inv)
        res[f"K7_{which}"] = [int((want7 != got7).sum()), int((want7 > 0).sum()), int((got7 > 0).sum())]
        got9, got_mask, got_src = K9.run(ref.update_uniforms(pose_t, ts), before, poses, frame, o_rc, o_idx)
        kept_g, kept_w = np.zeros(n, bool), np.zeros(n, bool)
        kept_g[got_src], kept_w[want_src] = True, True
        res[f"K9_{which}"] = [int((kept_g != kept_w).sum()), int(((got_mask > 0.5) != (want_mask[..., 0] > 0.5)).sum()), int(got9.shape[0])]
        got10 = K10.run(ref.generate_uniforms(pose_t, ts), frame, o_rc, mask4)
        res[f"K10_{which}"] = [int(got10.shape[0]), int(want10.shape[0]),
                               bool(got10.shape == want10.shape and np.array_equal(got10[:, 8:16].view(np.uint32), want10[:, 8:16].view(np.uint32)))]
    print(res)
    REPORT["passes_64x900"] = res
    assert res["K1_detmath"] == 0 and 0 < res["K1_driver"] < 0.02 * W * H
    # what is left is the last ulp of the driver's own mat * vec / dot (its fma policy is not this repository's)
    assert res["K7_detmath"][0] <= 3 and abs(res["K7_detmath"][1] - res["K7_detmath"][2]) <= 2 and res["K7_driver"][0] > 100
    assert res["K4_detmath"] >= 0.99 > res["K4_driver"] >= 0.97
    assert res["K9_detmath"][0] <= 2 and res["K9_detmath"][1] <= 3
    assert abs(res["K9_detmath"][2] - want9.view(np.float32).reshape(-1, 16).shape[0]) <= 2
    assert res["K10_detmath"][0] == res["K10_detmath"][1] and res["K10_detmath"][2]


def test_free_running_gl_pipeline_with_specified_transcendentals(gl, oracle_lib):
    """tests/test_gl_pipeline.py::test_free_running_gl_pipeline_against_the_oracle with the specified angle functions in
    every pass: eight scans (8.8 m), each side's maps and poses feed its own next scan.  The driver's asin accounted for
    most of the 1 - 2 cm of round 4: now the trajectories stay within 1 cm / 1e-3 rad (measured 9 mm / 9e-4 rad; what is
    left is K4's attribute interpolation, 0.5 % of the model texels, through ten gated iterations per scan)."""
    from oracle import glpipeline
    W = 900
    p = params_with_size(W)
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    with gl.transcendentals("detmath"):
        g = glpipeline.GLPipeline(p)
    rows = []
    for k in range(8):
        pts, lab, prob, _ = get_scan(k, W, True)
        op.process_scan(pts, lab, prob, fixed_iterations=ITER)
        g.process_scan(pts, lab, prob, ITER)
        dt, dr = pose_delta(op.pose(), g.current_pose)
        rows.append((k, dt, dr, op.ctx.map_size(), g.counts["map"]))
        print("scan %d: %.2e m %.2e rad | map %d / %d" % rows[-1])
        assert dt <= 1.5e-2 and dr <= 1.5e-3
        assert abs(rows[-1][3] - rows[-1][4]) <= 0.002 * rows[-1][3] + 5
    REPORT["free_running_64x900_specified"] = rows


def test_the_glsl_prelude_is_the_specified_arithmetic(gl, tmp_path):
    """oracle/glref.py::DETMATH_PRELUDE against include/suma_detmath.h itself (tests/detmath_shim.c compiles the header):
    asin / acos / atan / atan(y, x) evaluated by llvmpipe through a transform-feedback shader on 200 000 arguments are
    the header's values BIT FOR BIT -- including its fused multiply-add steps, which the prelude forms through double
    (llvmpipe lowers GLSL's own fma() to a multiply and an add: two roundings, measured).  Without this the control of
    the acceptance test would compare two different libraries.  The driver's own functions on the same arguments, for
    the record: asin up to 3.9e-4 rad away, acos 1.6e-4, atan 3e-6."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "detmath_shim.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off",
                           os.path.join(os.path.dirname(__file__), "detmath_shim.c"), "-o", so, "-lm"])
    shim = C.CDLL(so)
    rng = np.random.default_rng(11)
    n = 200_000
    x = np.concatenate([rng.uniform(-1, 1, n // 2), rng.uniform(-1e-3, 1e-3, n // 4), rng.normal(0, 30, n // 4)]).astype(np.float32)
    y = rng.normal(0, 20, n).astype(np.float32)
    xc = np.clip(x, -1, 1)
    vs = """#version 330
in float ax; in float ay; out vec4 r;
void main() { r = vec4(asin(clamp(ax, -1.0, 1.0)), acos(clamp(ax, -1.0, 1.0)), atan(ax), atan(ay, ax)); gl_Position = vec4(0.0); }"""
    got = gl.eval_vertex_shader(vs, {"ax": x, "ay": y}, 4, prelude="detmath")
    stock = gl.eval_vertex_shader(vs, {"ax": x, "ay": y}, 4, prelude="driver")
    want = np.zeros((n, 4), np.float32)
    for col, (name, arg) in enumerate((("t_asin", xc), ("t_acos", xc), ("t_atan", x))):
        out = np.empty(n, np.float32)
        getattr(shim, name)(np.ascontiguousarray(arg).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n)
        want[:, col] = out
    out = np.empty(n, np.float32)
    shim.t_atan2(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n)
    want[:, 3] = out
    ne = got.view(np.uint32) != want.view(np.uint32)
    assert not ne.any(), f"{int(ne.any(axis=1).sum())} of {n} arguments differ, columns {ne.sum(axis=0).tolist()}"
    err = np.abs(stock.astype(np.float64) - want).max(axis=0)
    print("llvmpipe's own asin / acos / atan / atan2: max abs difference from the specified functions", err)
    REPORT["driver_transcendentals_max_abs_err"] = dict(zip(("asin", "acos", "atan", "atan2"), map(float, err)))
    assert err[0] > 1e-4, "the driver's asin is what round 4 measured"
