"""Known-answer tests that pin the CPU oracle (oracle/) -- the reference ships no tests, golden vectors or
fixtures; besides the stage-by-stage comparison with the reference's own shaders compiled by g++
(tests/test_ref_shaders.py), the oracle is pinned by closed-form answers derived from
the reference's formulas, by invariants, and by an independent build with glibc transcendentals.
"""
import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import ACC_SCALE, params_with_size

W, H = 360, 32  # small images keep the CPU suite fast


def plane_scan(width=W, height=H, z0=-1.73, n_az=720, n_el=64):
    """dense rays onto the ground plane z = z0 (sensor at the origin)"""
    az = (np.arange(n_az) + 0.5) * (2 * np.pi / n_az) - np.pi
    el = np.deg2rad(np.linspace(-24.5, -2.5, n_el))
    A, E = np.meshgrid(az, el, indexing="ij")
    d = np.stack([np.cos(A) * np.cos(E), np.sin(A) * np.cos(E), np.sin(E)], -1).reshape(-1, 3)
    t = z0 / d[:, 2]
    pts = np.ones((d.shape[0], 4), np.float32)
    pts[:, :3] = d * t[:, None]
    return pts[(t > 2.5) & (t < 70)]


def test_k2_normals_of_a_plane(oracle_lib):
    p = params_with_size(W, H)
    ora = oracle_lib.Oracle(p)
    pts = plane_scan()
    f = ora.preprocess(pts, None, None, 20, ora.frame())
    n, v = f.normal.copy(), f.vertex.copy()
    ok = (n[..., 3] > 0.5) & (v[..., 3] > 0.5)
    assert ok.sum() > 0.3 * W * H
    # gen_normalmap.frag:41-99: cross(normalize(u-p), normalize(v-p)) of plane points = the plane normal
    assert np.abs(np.abs(n[ok][:, 2]) - 1.0).max() < 2e-3
    assert np.abs(n[ok][:, :2]).max() < 5e-2
    assert (n[ok][:, 2] > 0).all()  # faces the sensor (x to the right, y up in the image => +z)
    # invalid vertex pixels carry normal (0,0,0,1) and semantic (0,0,0,1) (quirk B-2)
    inv = v[..., 3] < 0.5
    assert (n[inv] == [0, 0, 0, 1]).all() and (f.semantic[inv] == [0, 0, 0, 1]).all()


def test_k8_radius_formula(oracle_lib):
    p = params_with_size(W, H)
    ora = oracle_lib.Oracle(p)
    f = ora.preprocess(plane_scan(), None, None, 20, ora.frame())
    ora.map_update(np.eye(4), f)
    rc = ora.map_radius_conf()
    v, n = f.vertex, f.normal
    d = np.linalg.norm(v[..., :3], axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        cosv = np.einsum("...k,...k", n[..., :3], -v[..., :3] / d[..., None])
    valid = (v[..., 3] > .5) & (n[..., 3] > .5) & (cosv > np.cos(np.deg2rad(90.0)))
    pixel_size = max(np.tan(0.5 * np.deg2rad(28.0) / H), np.tan(0.5 * 2 * np.pi / W))  # SurfelMap.cpp:339-344
    expect = np.clip(1.41 * d * pixel_size / np.clip(cosv, 0.5, 1.0), p.min_radius, p.max_radius)
    assert np.array_equal(rc[..., 3] > 0.5, valid)
    assert np.allclose(rc[..., 0][valid], expect[valid], rtol=2e-5)
    assert not rc[..., 1].any()  # quirk B-3
    # every valid, front-facing pixel of the first scan becomes a surfel
    assert ora.map_size() == int((valid & (cosv > 0.01)).sum()) == ora.map_counts()[1]


def test_k6_identity_gives_zero_gradient_and_closed_form_JtJ(oracle_lib):
    p = params_with_size(W, H, bilinear_sampling=0)
    ora = oracle_lib.Oracle(p)
    pts, lab, prob, _ = get_scan(3, W, False, H)
    f = ora.preprocess(pts, None, None, 20, ora.frame())
    F, acc, JtJ, Jtr, st = ora.jacobian_products(f, f, np.eye(4), 0)
    v, n = f.vertex.astype(np.float64), f.normal.astype(np.float64)
    ok = (v[..., 3] + n[..., 3]) > 1.5
    # residual n.(v - v) = 0 exactly -> weight 1, F = 0, Jtr = 0 (Frame2Model_jacobians.geom:109-116)
    assert F == 0.0 and not Jtr.any() and st.outlier == 0
    assert st.valid == int(ok.sum()) and st.valid + st.invalid == W * H and st.inlier == st.valid
    J = np.concatenate([n[ok][:, :3], np.cross(v[ok][:, :3], n[ok][:, :3])], axis=1)
    ref = J.T @ J
    assert np.allclose(JtJ, ref, rtol=1e-5, atol=1e-3 * np.abs(ref).max() * 1e-3)
    assert np.array_equal(JtJ, JtJ.T) and acc[29] == st.valid
    # fixed-point accumulation is exact: every word is an integer multiple of 2^-28 of the sum
    assert JtJ[0, 0] == acc[0] / ACC_SCALE


def test_gauss_newton_recovers_a_known_rigid_offset(oracle_lib):
    p = params_with_size(W, H, max_iterations=30)
    ora = oracle_lib.Oracle(p)
    pts, _, _, _ = get_scan(3, W, False, H)
    a = np.deg2rad(0.8)
    T = np.eye(4)
    T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    T[:3, 3] = [0.25, -0.08, 0.02]
    ptsB = pts.copy()
    ptsB[:, :3] = (pts[:, :3] - T[:3, 3]) @ T[:3, :3]  # the same world points seen from pose T
    fA = ora.preprocess(pts, None, None, 20, ora.frame())
    fB = ora.preprocess(ptsB, None, None, 20, ora.frame())
    Tout, hist, st = ora.minimize(fB, fA, np.eye(4))
    err = np.linalg.inv(T) @ Tout
    assert np.linalg.norm(err[:3, 3]) < 0.03, err[:3, 3]
    assert np.arccos(np.clip(0.5 * (np.trace(err[:3, :3]) - 1), -1, 1)) < np.deg2rad(0.15)
    assert hist.shape[0] == st.iterations + 1 and np.array_equal(hist[0], np.eye(4))
    # the mean weighted inlier residual drops
    s0 = ora.jacobian_products(fB, fA, np.eye(4), 0)[4]
    s1 = ora.jacobian_products(fB, fA, Tout, 0)[4]
    assert s1.inlier_residual / s1.inlier < 0.5 * s0.inlier_residual / s0.inlier


def test_se3_exp_and_ldlt(oracle_lib):
    assert np.array_equal(oracle_lib.se3_exp(np.zeros(6)), np.eye(4))
    th = 0.3
    T = oracle_lib.se3_exp([0, 0, 0, 0, 0, th])
    assert np.allclose(T[:2, :2], [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], atol=1e-15)
    T = oracle_lib.se3_exp([1, 2, 3, 0, 0, 0])
    assert np.array_equal(T[:3, 3], [1, 2, 3]) and np.array_equal(T[:3, :3], np.eye(3))
    # pure screw: translation along the axis is preserved, V = I + ... (lie_algebra.cpp:4-34)
    T = oracle_lib.se3_exp([0, 0, 0.5, 0, 0, 0.7])
    assert abs(T[2, 3] - 0.5) < 1e-15
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6))
    A = A @ A.T + 6 * np.eye(6)
    b = rng.normal(size=6)
    assert np.allclose(oracle_lib.solve6(A, b), np.linalg.solve(A, -b), rtol=1e-12)  # JtJ.ldlt().solve(-Jtf)


def test_pipeline_invariants_and_libm_cross_check(oracle_lib):
    """three scans through processScan: structural invariants + agreement of the deterministic-math
    oracle with the same restatement built on glibc's transcendentals"""
    p = params_with_size(W, H)
    a = oracle_lib.OraclePipeline(p)
    b = oracle_lib.OraclePipeline(p, variant="libm")
    sizes = []
    for k in range(3):
        pts, lab, prob, _ = get_scan(k, W, True, H)
        for q in (a, b):
            q.process_scan(pts, lab, prob, fixed_iterations=8)
        s = a.ctx.map_surfels()
        su, sn = a.ctx.map_counts()
        assert len(s) == su + sn  # K11 keeps everything this close to the origin
        assert (s["timestamp"] <= k).all() and (s["count"] <= k).all() and (s["radius"] > 0).all()
        assert np.allclose(np.linalg.norm(np.stack([s["nx"], s["ny"], s["nz"]], 1), axis=1), 1, atol=1e-4)
        new = s[su:]
        assert (new["count"] == k).all() and (new["timestamp"] == k).all() and (new["weight"] == 1).all()
        if k:  # transform feedback order: survivors keep their relative order => creation stamps are sorted
            assert (np.diff(s[:su]["count"]) >= 0).all()
        st = a.last_stats()
        assert st.valid + st.invalid == W * H or k == 0
        sizes.append(len(s))
        assert abs(len(s) - b.ctx.map_size()) <= 0.01 * len(s) + 5
        # last-ulp differences of the transcendentals move a few pairs across a gate (tests/test_gl_controls.py measures the
        # same effect with a real GL's asin): millimetres on this small image, never more
        assert np.abs(a.pose() - b.pose()).max() < 3e-3
    assert sizes[0] < sizes[1] < sizes[2]
    assert 0.3 < a.pose()[0, 3] / 2.2 < 1.5  # two steps of 1.1 m along x


def test_label_offset_quirk(oracle_lib):
    """quirk B-1 (Preprocessing.cpp:142-145): point i receives labels[i+4] / probs[i+5]"""
    pts = np.array([[10, 0, 0, 1], [0, 10, 0, 1], [-10, 0, 0, 1], [0, -10, 0, 1]] * 3, np.float32)
    pts[4:, 0] += 1000  # out of range: never drawn
    lab = np.arange(12, dtype=np.float32) * 10 + 40
    prob = np.linspace(0.1, 0.9, 12).astype(np.float32)
    for off, expect in (((4, 5), lab[4]), ((0, 0), lab[0])):
        p = params_with_size(W, H, label_offset=off[0], prob_offset=off[1])
        ora = oracle_lib.Oracle(p)
        f = ora.preprocess(pts, lab, prob, 20, ora.frame())
        sem = f.semantic
        hit = np.argwhere(f.vertex[..., 0] == 10.0)
        assert len(hit) == 1
        assert sem[tuple(hit[0])][0] == np.float32(expect) / np.float32(255.0)


def test_oracle_threads_are_bit_identical(oracle_lib):
    """ora_set_threads: integer sums, z-buffer minima and stable compactions do not depend on the thread count."""
    from semantic_suma_amd import synth
    p = params_with_size(360, height=32)
    res = []
    for th in (1, 4):
        op = oracle_lib.OraclePipeline(p, threads=th)
        for k in range(4):
            pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=360, height=32)
            op.process_scan(pts, lab, prob, fixed_iterations=5)
        res.append((op.pose().tobytes(), op.ctx.map_surfels().tobytes(), op.last_stats().as_dict()))
    assert res[0] == res[1]


def test_native_build_is_bit_identical(oracle_lib, scans):
    """bench.py times the oracle built -O3 -march=native; it must be the same function as the -O2
    library the parity tests use: -ffp-contract=off keeps FMA-capable hosts from fusing, integer sums are exact"""
    if not oracle_lib.build_native():
        pytest.skip("no compiler for the native build")
    from semantic_suma_amd.types import params_with_size
    p = params_with_size(360, 32)
    a, b = oracle_lib.OraclePipeline(p), oracle_lib.OraclePipeline(p, variant="native", threads=4)
    for k in range(5):
        pts, lab, prob, _ = scans(k, 360, True, 32)
        a.process_scan(pts, lab, prob, fixed_iterations=6)
        b.process_scan(pts, lab, prob, fixed_iterations=6)
        assert np.array_equal(a.pose(), b.pose()), f"scan {k}"
    assert a.ctx.map_surfels().tobytes() == b.ctx.map_surfels().tobytes()
    assert a.last_stats().as_dict() == b.last_stats().as_dict()


def test_vertexmap_filters_known_answers(oracle_lib):
    """Preprocessing.cpp:150-236 on hand-made inputs: the blended K1 sums a texel's points in index order (fp32, not
    associative), avg_vertexmap.frag divides by the count, the bilateral filter on a constant-range patch;
    and the texel a filter fragment addresses, int(((x + 1/2) / W) * W), is x for every image width up to 8192."""
    for Wd in range(1, 8193):
        px = np.arange(Wd, dtype=np.float32)
        got = (((px + np.float32(0.5)) / np.float32(Wd)) * np.float32(Wd)).astype(np.int32)
        assert np.array_equal(got, np.arange(Wd)), Wd
    W, H = 64, 16
    # three points in one texel whose x sum depends on the order: (1e8 + 1) - 1e8 = 0 in fp32, 1e8 - 1e8 + 1 = 1
    pts = np.array([[10.0, 0, 0, 1], [10.0000095, 0, 0, 1], [9.999999, 0, 0, 1]], np.float32)
    p = params_with_size(W, H, avg_vertexmap=1, filter_sampling=1)
    ora = oracle_lib.Oracle(p)
    f = ora.preprocess(pts, None, None, 12, ora.frame())
    v = f.vertex.reshape(H, W, 4)
    hit = np.argwhere(v[..., 3] > 0.5)
    assert len(hit) == 1
    y, x = hit[0]
    s = np.float32(0)
    for q in pts[:, 0]:
        s = np.float32(q + s)
    assert v[y, x, 0] == np.float32(s / np.float32(3.0)) and v[y, x, 3] == 1.0
    # GL initial state (LINEAR, CLAMP_TO_EDGE): the integer coordinate is a texel corner, so the sum spreads over the
    # four texels that touch it with weight 1/4 each: count 3/4 > 0.5, each of them is "valid" and gets the mean
    ora2 = oracle_lib.Oracle(params_with_size(W, H, avg_vertexmap=1))
    f2 = ora2.preprocess(pts, None, None, 12, ora2.frame())
    v2 = f2.vertex.reshape(H, W, 4)
    q = np.float32(0.25)
    assert np.count_nonzero(v2[..., 3] > 0) == 4 and {tuple(h) for h in np.argwhere(v2[..., 3] > 0).tolist()} == \
        {(y, x), (y + 1, x), (y, x + 1), (y + 1, x + 1)}
    assert v2[y + 1, x + 1, 3] == 1.0 and v2[y + 1, x + 1, 0] == np.float32(np.float32(s * q) / np.float32(np.float32(3) * q))
    # bilateral filter on a sphere of constant range 8: every neighbour's "range" is length(vec4) = sqrt(8^2 + 1^2),
    # w included (bilateral_filter.frag:64), so the filtered range is sqrt(65) whatever the weights are
    az = (np.arange(W) + 0.5) / W * 2 * np.pi - np.pi
    el = np.deg2rad(np.linspace(-24.5, 2.5, H))
    A, E = np.meshgrid(az, el)
    sph = np.stack([8 * np.cos(A) * np.cos(E), 8 * np.sin(A) * np.cos(E), 8 * np.sin(E), np.ones_like(A)], -1)
    sph = sph.reshape(-1, 4).astype(np.float32)
    base = oracle_lib.Oracle(params_with_size(W, H))
    f0 = base.preprocess(sph, None, None, 12, base.frame())
    v0 = f0.vertex.reshape(H, W, 4).copy()
    bf = oracle_lib.Oracle(params_with_size(W, H, filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5,
                                            filter_sampling=1))
    f1 = bf.preprocess(sph, None, None, 12, bf.frame())
    v1 = f1.vertex.reshape(H, W, 4).copy()
    valid = v0[..., 3] > 0.5
    assert valid.mean() > 0.9 and np.array_equal(v1[..., 3] > 0.5, valid)
    assert np.abs(v1[valid][:, :3] - v0[valid][:, :3] * np.float32(np.sqrt(65.0) / 8.0)).max() < 2e-5
    # ... and with use_filtered_vertexmap off the filter's result is dropped (Preprocessing.cpp:234)
    off = oracle_lib.Oracle(params_with_size(W, H, filter_vertexmap=1, use_filtered_vertexmap=0, bilateral_sigma_space=4.5))
    f3 = off.preprocess(sph, None, None, 12, off.frame())
    assert np.array_equal(f3.vertex.reshape(H, W, 4), v0)
