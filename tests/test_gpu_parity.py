"""GPU parity tests (run with -m gpu on an MI355X): the hand-written gfx950 path, called through
the C-ABI (libsuma_hip.so), against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact integer outputs (index map, integration mask, surfel
counts / order / timestamps); pose delta <= 1e-4 m / 1e-5 rad per ICP iteration.  Because oracle
and kernels evaluate the same IEEE binary32 operation sequence (include/suma_detmath.h, no FMA
contraction) and the J^T J sums are exact fixed point, every float output is in fact compared
BIT FOR BIT here; the tolerances appear only where stated.
"""
import os

import numpy as np
import pytest

from conftest import assert_bit_equal, get_scan
from semantic_suma_amd.types import params_with_size

pytestmark = pytest.mark.gpu
THREADS = max(1, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def hip():
    from semantic_suma_amd import core
    core.lib()  # raises if libsuma_hip.so is missing: no silent fallback
    return core


def frames_equal(hf, of, what):
    for m, name in enumerate(("vertex", "normal", "semantic")):
        assert_bit_equal(hf.download(m), of.map(m), f"{what}.{name}")


def pose_delta(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    t = float(np.linalg.norm(d[:3, 3]))
    r = float(np.arccos(np.clip(0.5 * (np.trace(d[:3, :3]) - 1.0), -1.0, 1.0)))
    return t, r


@pytest.mark.parametrize("width,semantics", [(900, False), (900, True), (2048, True)])
def test_preprocess_k1_k3(hip, oracle_lib, width, semantics):
    p = params_with_size(width)
    pts, lab, prob, _ = get_scan(0, width, semantics)
    ctx = hip.Context(p)
    ora = oracle_lib.Oracle(p)
    for timestamp in (0, 12):  # < 10 drops dynamic classes (gen_vertexmap.vert:95-102)
        hf = hip.Frame(ctx, width, 64)
        hip.Preprocessing(ctx).process(pts, hf, lab, prob, timestamp)
        of = ora.preprocess(pts, lab, prob, timestamp, ora.frame())
        frames_equal(hf, of, f"preprocess w={width} t={timestamp}")
        v = hf.download(0)
        assert 0.5 < float((v[..., 3] > 0).mean()) <= 1.0  # the synthetic scan fills most of the image


FILTER_VARIANTS = [
    dict(avg_vertexmap=1),
    dict(avg_vertexmap=1, filter_sampling=1),
    dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5, bilateral_sigma_range=2.5),
    dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5, bilateral_sigma_range=2.5, filter_sampling=1),
    dict(avg_vertexmap=1, filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=2.0, bilateral_sigma_range=0.5),
    dict(filter_vertexmap=1, use_filtered_vertexmap=0, bilateral_sigma_space=4.5),  # computed and dropped (Preprocessing.cpp:234)
]


@pytest.mark.parametrize("overrides", FILTER_VARIANTS, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_preprocess_vertexmap_filters(hip, oracle_lib, overrides):
    """Preprocessing.cpp:150-236: blended K1 + avg_vertexmap.frag, bilateral_filter.frag, both texture states; a dense
    cloud (three scans stacked: ~3 points per texel, in three different orders) so that the ordered sums have terms"""
    from semantic_suma_amd import synth
    W, H = 2048, 64
    p = params_with_size(W, H, **overrides)
    ctx, ora = hip.Context(p), oracle_lib.Oracle(p, threads=THREADS)
    a = synth.generate_scan(0, n_azimuth=W)
    b = synth.generate_scan(0, n_azimuth=W, seed=99)
    c = synth.generate_scan(0, n_azimuth=W, seed=7)
    pts = np.concatenate([a[0], b[0][::-1], c[0]]).astype(np.float32)
    lab = np.concatenate([a[1], b[1][::-1], c[1]]).astype(np.float32)
    prob = np.concatenate([a[2], b[2][::-1], c[2]]).astype(np.float32)
    for timestamp in (0, 12):
        hf = hip.Frame(ctx, W, H)
        hip.Preprocessing(ctx).process(pts, hf, lab, prob, timestamp)
        of = ora.preprocess(pts, lab, prob, timestamp, ora.frame())
        frames_equal(hf, of, f"filters {overrides} t={timestamp}")
        assert float((hf.download(0)[..., 3] > 0.5).mean()) > 0.5
    if overrides.get("avg_vertexmap"):
        # pathological: every point in one texel (one run of n terms), plus clipped and NaN points
        one = np.tile(np.array([[10.0, 0.0, 0.0, 1.0]], np.float32), (5000, 1))
        one[:, 2] = np.linspace(-0.001, 0.001, 5000, dtype=np.float32)
        one = np.concatenate([one, np.array([[0, 0, 0, 1], [np.nan, 1, 1, 1], [500, 0, 0, 1]], np.float32)])
        hf = hip.Frame(ctx, W, H)
        hip.Preprocessing(ctx).process(one, hf, None, None, 12)
        of = ora.preprocess(one, None, None, 12, ora.frame())
        frames_equal(hf, of, "all points in one texel")
        empty = np.zeros((0, 4), np.float32)
        hip.Preprocessing(ctx).process(empty, hf, None, None, 12)
        of = ora.preprocess(empty, None, None, 12, ora.frame())
        frames_equal(hf, of, "empty scan, filters on")


def test_pipeline_with_vertexmap_filters(hip, oracle_lib):
    W = 900
    for overrides in (dict(avg_vertexmap=1, filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=3.0,
                           bilateral_sigma_range=1.0, filter_sampling=1),
                      dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5)):
        p = params_with_size(W, **overrides)
        hp, op = hip.SurfelMapping(p), oracle_lib.OraclePipeline(p, threads=THREADS)
        for k in range(5):
            pts, lab, prob, _ = get_scan(k, W, True)
            hp.processScan(pts, lab, prob, fixed_iterations=6)
            op.process_scan(pts, lab, prob, fixed_iterations=6)
            assert np.array_equal(hp.getCurrentPose(), op.pose()), f"{overrides} scan {k}: pose"
            assert hp.lastStats().as_dict() == op.last_stats().as_dict()
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"{overrides}: map"
        assert hp.map.size() > 20000


def test_vertexmap_filters_switched_on_a_live_context(hip, oracle_lib):
    """setParameters on a context that was created without the filters: their buffers are allocated on first use"""
    W = 900
    pts, lab, prob, _ = get_scan(3, W, True)
    ctx = hip.Context(params_with_size(W))
    for ov in (dict(), dict(avg_vertexmap=1, filter_sampling=1),
               dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=3.0), dict()):
        p = params_with_size(W, **ov)
        ctx.set_params(p)
        ora = oracle_lib.Oracle(p, threads=THREADS)
        hf = hip.Frame(ctx, W, 64)
        hip.Preprocessing(ctx).process(pts, hf, lab, prob, 12)
        of = ora.preprocess(pts, lab, prob, 12, ora.frame())
        frames_equal(hf, of, f"live switch {ov}")


def test_vertexmap_filter_parameter_errors(hip):
    with pytest.raises(hip.SumaError, match="bilateral_sigma_space"):
        hip.Context(params_with_size(900, filter_vertexmap=1))  # default.xml holds no bilateral_sigma_space
    with pytest.raises(hip.SumaError, match="filter_sampling"):
        hip.Context(params_with_size(900, filter_sampling=7))
    ctx = hip.Context(params_with_size(900))
    with pytest.raises(hip.SumaError, match="bilateral_sigma_space"):
        ctx.set_params(params_with_size(900, filter_vertexmap=1))
    # the rasteriser's fixed-point window coordinates (|X| < 2^21 in 1/256 pixel) bound the model width; the range is
    # enforced, not assumed (round-5 advisor)
    wide = params_with_size(900)
    wide.model_width = 5462
    with pytest.raises(hip.SumaError, match="model_width"):
        hip.Context(wide)
    with pytest.raises(hip.SumaError, match="fixed at creation"):  # and a context's image sizes cannot be changed later
        ctx.set_params(wide)


def test_preprocess_edge_cases(hip, oracle_lib):
    p = params_with_size(900)
    ctx = hip.Context(p)
    ora = oracle_lib.Oracle(p)
    # empty scan
    hf = hip.Frame(ctx, 900, 64)
    empty = np.zeros((0, 4), dtype=np.float32)
    hip.Preprocessing(ctx).process(empty, hf, None, None, 0)
    of = ora.preprocess(empty, None, None, 0, ora.frame())
    frames_equal(hf, of, "empty scan")
    assert not hf.download(0).any()
    assert (hf.download(2)[..., 3] == 1.0).all()  # quirk B-2: invalid pixels carry semantic (0,0,0,1)
    # degenerate points: origin, out of range, out of FOV, NaN, duplicates competing for one pixel
    pts = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [100, 0, 0, 1], [0, 0, 5, 1], [np.nan, 1, 1, 1],
                    [10, 0, 0, 1], [10, 0, 0, 1], [10.0000005, 0, 0, 1], [5, 5, -1, 1], [-7, 0.001, -0.5, 1]],
                   dtype=np.float32)
    lab = np.arange(10, dtype=np.float32) * 10
    prob = np.linspace(0.1, 1.0, 10).astype(np.float32)
    hip.Preprocessing(ctx).process(pts, hf, lab, prob, 3)
    of = ora.preprocess(pts, lab, prob, 3, ora.frame())
    frames_equal(hf, of, "degenerate points")
    # no semantics: NULL labels / probs
    pts2, _, _, _ = get_scan(1, 900, False)
    hip.Preprocessing(ctx).process(pts2, hf, None, None, 20)
    of = ora.preprocess(pts2, None, None, 20, ora.frame())
    frames_equal(hf, of, "null labels")


def _model_and_data(oracle_lib, p, width, semantics, k_model=0, k_data=1):
    """oracle-made model frame (rendered from a one-scan map) and data frame"""
    ora = oracle_lib.Oracle(p)
    pts0, l0, p0, _ = get_scan(k_model, width, semantics)
    pts1, l1, p1, _ = get_scan(k_data, width, semantics)
    f0 = ora.preprocess(pts0, l0, p0, 20, ora.frame())
    f1 = ora.preprocess(pts1, l1, p1, 21, ora.frame())
    return ora, f0, f1


@pytest.mark.parametrize("width,semantics,weight", [(900, False, 1), (900, True, 1), (900, True, 2), (2048, True, 1)])
def test_k6_jacobian_products(hip, oracle_lib, width, semantics, weight):
    p = params_with_size(width, weight_function=weight)
    ora, f0, f1 = _model_and_data(oracle_lib, p, width, semantics)
    ctx = hip.Context(p)
    h0, h1 = hip.Frame(ctx, width, 64), hip.Frame(ctx, width, 64)
    h0.set(f0.vertex, f0.normal, f0.semantic)
    h1.set(f1.vertex, f1.normal, f1.semantic)
    obj = hip.Frame2Model(ctx)
    obj.setData(h1, h0)
    T = np.eye(4)
    T[:3, 3] = [1.05, 0.02, 0.0]
    for iteration in (0, 1):
        obj.initialize(T)
        obj._iteration = iteration
        F, JtJ, Jtr = obj.jacobianProducts()
        Fo, acc, JtJo, Jtro, st = ora.jacobian_products(f1, f0, T, iteration)
        np.testing.assert_array_equal(obj.acc, acc)  # exact fixed-point sums
        assert F == Fo and np.array_equal(JtJ, JtJo) and np.array_equal(Jtr, Jtro)
        assert (obj.valid(), obj.outlier(), obj.inlier(), obj.invalid()) == (st.valid, st.outlier, st.inlier, st.invalid)
        assert obj.valid() + obj.invalid() == width * 64
        assert obj.valid() > 1000


def test_k6_nearest_sampling_and_empty(hip, oracle_lib):
    p = params_with_size(900, bilinear_sampling=0)
    ora, f0, f1 = _model_and_data(oracle_lib, p, 900, True)
    ctx = hip.Context(p)
    h0, h1 = hip.Frame(ctx, 900, 64), hip.Frame(ctx, 900, 64)
    h0.set(f0.vertex, f0.normal, f0.semantic)
    h1.set(f1.vertex, f1.normal, f1.semantic)
    obj = hip.Frame2Model(ctx)
    obj.setData(h1, h0)
    obj.initialize(np.eye(4))
    F, JtJ, Jtr = obj.jacobianProducts()
    Fo, acc, JtJo, Jtro, st = ora.jacobian_products(f1, f0, np.eye(4), 0)
    np.testing.assert_array_equal(obj.acc, acc)
    # all-invalid model: nothing associates
    z = np.zeros((64, 900, 4), dtype=np.float32)
    h0.set(z, z, z)
    obj.initialize(np.eye(4))
    F, JtJ, Jtr = obj.jacobianProducts()
    assert F == 0.0 and not JtJ.any() and obj.valid() == 0 and obj.invalid() == 900 * 64


@pytest.mark.parametrize("width", [900, 2048])
def test_gauss_newton_minimize(hip, oracle_lib, width):
    p = params_with_size(width, max_iterations=10)
    ora, f0, f1 = _model_and_data(oracle_lib, p, width, True)
    ctx = hip.Context(p)
    h0, h1 = hip.Frame(ctx, width, 64), hip.Frame(ctx, width, 64)
    h0.set(f0.vertex, f0.normal, f0.semantic)
    h1.set(f1.vertex, f1.normal, f1.semantic)
    obj = hip.Frame2Model(ctx)
    obj.setData(h1, h0)
    gn = hip.LieGaussNewton(ctx)
    gn.minimize(obj, np.eye(4))
    To, hist_o, st = ora.minimize(f1, f0, np.eye(4))
    assert gn.history().shape == hist_o.shape
    # north_star bar: <= 1e-4 m / 1e-5 rad per ICP iteration (the implementation is in fact bit-exact)
    for k in range(hist_o.shape[0]):
        t, r = pose_delta(gn.history()[k], hist_o[k])
        assert t <= 1e-4 and r <= 1e-5, f"iteration {k}: {t} m, {r} rad"
    assert np.array_equal(gn.pose(), To), "final pose differs in bits"
    assert gn.iterationCount() == st.iterations and gn.stats.converged == st.converged
    # the scan pair is 1.1 m apart along x: ICP must have moved most of the way
    assert 0.8 < gn.pose()[0, 3] < 1.4


def test_gauss_newton_convergence_and_batch(hip, oracle_lib):
    p = params_with_size(900)  # default.xml: 33 iterations, eps = delta = 1e-4 -> stops early
    ora, f0, f1 = _model_and_data(oracle_lib, p, 900, False)
    ctx = hip.Context(p)
    h0, h1 = hip.Frame(ctx, 900, 64), hip.Frame(ctx, 900, 64)
    h0.set(f0.vertex, f0.normal, f0.semantic)
    h1.set(f1.vertex, f1.normal, f1.semantic)
    obj = hip.Frame2Model(ctx)
    obj.setData(h1, h0)
    gn = hip.LieGaussNewton(ctx)
    rng = np.random.default_rng(1234)
    T0s = []
    for k in range(8):  # BASELINE config 3: 8 perturbed starts
        T = np.eye(4)
        T[:3, 3] = [1.0, 0, 0]
        if k:
            T[:3, 3] += rng.uniform(-0.2, 0.2, 3)
            a = np.deg2rad(rng.uniform(-2, 2))
            T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        T0s.append(T)
    Ts, stats = gn.minimize_batch(T0s, obj)
    for k in range(8):
        To, _, st = ora.minimize(f1, f0, T0s[k])
        assert np.array_equal(Ts[k], To), f"hypothesis {k}"
        assert stats[k]["iterations"] == st.iterations and stats[k]["converged"] == st.converged
        assert stats[k]["valid"] == st.valid and stats[k]["error"] == st.error
    gn.minimize(obj, T0s[0])
    assert np.array_equal(gn.pose(), Ts[0])
    # loose thresholds: the stopping tests of LieGaussNewton.cpp:64-66 must fire at the same iteration,
    # and the converged step is still applied (quirk B-4)
    p2 = params_with_size(900, stopping_threshold=0.5, delta=5e-3)  # (2e-3 under the round-4 arithmetic: the steps of this pair settle around 3 mm)
    ctx.set_params(p2)
    ora.set_params(p2)
    gn.minimize(obj, T0s[0])
    To, hist_o, st = ora.minimize(f1, f0, T0s[0])
    assert st.converged == 1 and st.iterations < 33, "test setup: the oracle should converge early"
    assert np.array_equal(gn.pose(), To) and gn.history().shape == hist_o.shape
    assert gn.stats.converged == 1 and gn.iterationCount() == st.iterations


def _run_maps(hip, oracle_lib, p, width, n_scans, semantics=True, step=1):
    """drive SurfelMap directly with ground-truth poses; compare every stage after every scan"""
    ctx = hip.Context(p)
    ora = oracle_lib.Oracle(p)
    hmap = hip.SurfelMap(ctx)
    hpre = hip.Preprocessing(ctx)
    T_first = None
    for i in range(n_scans):
        k = i * step
        pts, lab, prob, T = get_scan(k, width, semantics)
        if T_first is None:
            T_first = T
        pose = np.linalg.inv(T_first) @ T
        hf = hip.Frame(ctx, width, 64)
        hpre.process(pts, hf, lab, prob, i)
        of = ora.preprocess(pts, lab, prob, i, ora.frame())
        hmap.update(pose, hf)
        ora.map_update(pose, of)
        assert_bit_equal(hmap.radius_conf(), ora.map_radius_conf(), f"scan {i} radius_conf")
        np.testing.assert_array_equal(hmap.index_map(), ora.map_index_map(), err_msg=f"scan {i} index map")
        np.testing.assert_array_equal(hmap.integrated(), ora.map_integrated(), err_msg=f"scan {i} integration mask")
        su, sn, cached, origin = hmap.counts()
        assert (su, sn) == ora.map_counts(), f"scan {i} (S', D)"
        assert cached == ora.map_cached_surfels() and origin == ora.map_submap_origin()
        assert hmap.size() == ora.map_size(), f"scan {i} map size"
        hs, os_ = hmap.getAllSurfels(), ora.map_surfels()
        assert hs.tobytes() == os_.tobytes(), f"scan {i}: surfel buffers differ"
        # render from the new pose
        ct = -2.0 if i < 3 else 0.0
        hout = hip.Frame(ctx, p.model_width, p.model_height)
        oout = ora.frame(model=True)
        hmap.render(pose, pose, hout, ct)
        ora.map_render(pose, pose, ct, oout)
        frames_equal(hout, oout, f"scan {i} render out")
        frames_equal(hmap.oldMapFrame(), ora.map_frame(0), f"scan {i} old frame")
        frames_equal(hmap.newMapFrame(), ora.map_frame(1), f"scan {i} new frame")
    return ctx, ora, hmap


def test_surfel_map_update_and_render_900(hip, oracle_lib):
    _run_maps(hip, oracle_lib, params_with_size(900), 900, 5)


def test_surfel_map_update_and_render_2048(hip, oracle_lib):
    _run_maps(hip, oracle_lib, params_with_size(2048), 2048, 3)


def test_surfel_map_update_poses(hip, oracle_lib):
    """SurfelMap::updatePoses (SurfelMap.cpp:486-490): after a pose-graph optimisation the map is deformed by
    re-uploading the pose table only; rendering and the next update then see the surfels at their new places"""
    p = params_with_size(900)
    n = 4
    ctx, ora, hmap = _run_maps(hip, oracle_lib, p, 900, n)
    T0 = get_scan(0, 900, True)[3]
    poses = []
    for i in range(n):
        T = np.linalg.inv(T0) @ get_scan(i, 900, True)[3]
        D = np.eye(4)  # a smooth "optimised" correction growing along the trajectory
        a = 0.002 * i
        D[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        D[:3, 3] = [0.01 * i, -0.005 * i, 0.002 * i]
        poses.append((D @ T).astype(np.float32))
    hmap.updatePoses(poses)
    ora.map_update_poses(np.stack(poses))
    pose = poses[-1]
    hout = hip.Frame(ctx, p.model_width, p.model_height)
    oout = ora.frame(model=True)
    hmap.render(pose, pose, hout, 0.0)
    ora.map_render(pose, pose, 0.0, oout)
    frames_equal(hout, oout, "render after updatePoses")
    # ... and the next scan is fused into the deformed map
    pts, lab, prob, T = get_scan(n, 900, True)
    hf = hip.Frame(ctx, 900, 64)
    hip.Preprocessing(ctx).process(pts, hf, lab, prob, n)
    of = ora.preprocess(pts, lab, prob, n, ora.frame())
    pose_n = (np.linalg.inv(T0) @ T)
    hmap.update(pose_n, hf)
    ora.map_update(pose_n, of)
    assert hmap.getAllSurfels().tobytes() == ora.map_surfels().tobytes()


def test_pipeline_device_scans_equal_host_scans(hip):
    """suma_pipeline_process_scan_device (scans already in HBM) == suma_pipeline_process_scan (host pointers)"""
    p = params_with_size(900)
    a, b = hip.SurfelMapping(p), hip.SurfelMapping(p)
    for k in range(3):
        pts, lab, prob, _ = get_scan(k, 900, True)
        a.processScan(pts, lab, prob, fixed_iterations=5)
        d = (b.ctx.device_array(pts), b.ctx.device_array(lab), b.ctx.device_array(prob))
        b.processScanDevice(*d, pts.shape[0], fixed_iterations=5)
        assert np.array_equal(a.getCurrentPose(), b.getCurrentPose())
        assert a.map.getAllSurfels().tobytes() == b.map.getAllSurfels().tobytes()


def test_surfel_map_render_variants(hip, oracle_lib):
    p = params_with_size(900)
    ctx, ora, hmap = _run_maps(hip, oracle_lib, p, 900, 3)
    pts, lab, prob, T = get_scan(2, 900, True)
    pose = np.linalg.inv(get_scan(0, 900, True)[3]) @ T
    pose2 = pose.copy()
    pose2[:3, 3] += [0.3, -0.1, 0.02]
    # make some surfels "old": pretend the map is 150 scans further (timestamp threshold = t - 100)
    surf = hmap.getAllSurfels()
    hmap.upload(surf, 150)
    ora.map_upload(surf, 150)
    hmap.render_active(pose2, -1.0)
    ora.map_render_active(pose2, -1.0)
    hmap.render_inactive(pose, -1.0)
    ora.map_render_inactive(pose, -1.0)
    hmap.render_composed(pose, pose2, -1.0)
    ora.map_render_composed(pose, pose2, -1.0)
    for w, nm in ((0, "old"), (1, "new"), (2, "composed")):
        hf, of = hmap._frame(w), ora.map_frame(w)
        assert_bit_equal(hf.download(0), of.map(0), f"{nm}.vertex")
        assert_bit_equal(hf.download(1), of.map(1), f"{nm}.normal")
    assert hmap.oldMapFrame().download(0)[..., 3].sum() > 1000  # old surfels were rendered
    # non-compose mode
    p2 = params_with_size(900, compose_rendering=0)
    ctx.set_params(p2)
    ora.set_params(p2)
    hout, oout = hip.Frame(ctx, 900, 64), ora.frame(model=True)
    hmap.render(pose, pose2, hout, -1.0)
    ora.map_render(pose, pose2, -1.0, oout)
    frames_equal(hout, oout, "non-compose render")


def test_submap_paging(hip, oracle_lib):
    # fast traverse (5 scans apart = 5.5 m per step) with small submaps so that tiles shift, get
    # extracted to the cache and come back (SurfelMap.cpp:744-824)
    p = params_with_size(900, submap_extent=4.0, submap_dimension=2)
    ctx, ora, hmap = _run_maps(hip, oracle_lib, p, 900, 8, step=5)
    assert ora.map_submap_origin() != (0, 0)
    assert ora.map_cached_surfels() > 0


@pytest.mark.parametrize("width,n_scans", [(900, 6), (2048, 4)])
def test_pipeline_process_scan(hip, oracle_lib, width, n_scans):
    """SurfelMapping::processScan end to end: poses, stats, maps after every scan."""
    p = params_with_size(width)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    for k in range(n_scans):
        pts, lab, prob, T = get_scan(k, width, True)
        hp.processScan(pts, lab, prob, fixed_iterations=10)
        op.process_scan(pts, lab, prob, fixed_iterations=10)
        t, r = pose_delta(hp.getCurrentPose(), op.pose())
        assert t <= 1e-4 and r <= 1e-5, f"scan {k}: pose differs by {t} m / {r} rad"
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose bits"
        assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k} stats"
        assert hp.map.size() == op.ctx.map_size(), f"scan {k} map size"
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k} surfels"
        for w in (0, 1, 2):
            frames_equal(hp.frame(w), op.frame(w), f"scan {k} frame {w}")
    # odometry sanity against the synthetic ground truth (1.1 m per scan)
    gt = np.linalg.inv(get_scan(0, width, True)[3]) @ get_scan(n_scans - 1, width, True)[3]
    t, r = pose_delta(hp.getCurrentPose(), gt)
    assert t < 0.5, f"drift {t} m after {n_scans} scans"


def test_pipeline_fallback_icp(hip, oracle_lib):
    """A jump in the motion (two scans skipped) trips the frame-to-frame fallback minimisation
    (SurfelMapping.cpp:434-449): same decision, same poses, same maps as the oracle."""
    p = params_with_size(900)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    for n, k in enumerate([0, 1, 2, 3, 6, 7]):
        pts, lab, prob, _ = get_scan(k, 900, True)
        hp.processScan(pts, lab, prob, fixed_iterations=10)
        op.process_scan(pts, lab, prob, fixed_iterations=10)
        assert hp.trackLoss() == op.track_loss(), f"scan {n}: fallback decision"
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {n}: pose bits"
        assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {n} stats"
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {n} surfels"
        for w in (0, 1, 2):
            frames_equal(hp.frame(w), op.frame(w), f"scan {n} frame {w}")
    assert op.track_loss() >= 1, "the sequence was meant to trip the fallback"


@pytest.mark.parametrize("overrides", [
    dict(weighting_scheme=1, averaging_scheme=0),                  # running weights, averaged position
    dict(weighting_scheme=2, averaging_scheme=1, update_always=1), # view-angle weights, push along the normal
    dict(confidence_mode=0),                                       # constant p_stable
    dict(confidence_mode=1),                                       # angle term only
    dict(confidence_mode=2, weight_function=2),                    # distance term only, Tukey ICP weights
    dict(use_stability=0),                                         # no stability gating in K4 / K9
    dict(partial_extraction=0),                                    # submap extraction in one go
    dict(initialize_identity=1),                                   # ICP starts from identity, not the last increment
    dict(compose_rendering=0),                                     # SurfelMap::render without the old/new composition
    dict(fallback_mode=0, icp_max_distance=1.0, icp_max_angle=20.0),
], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_pipeline_update_variants(hip, oracle_lib, overrides):
    """the switches of update_surfels.vert / SurfelMap.cpp that default.xml leaves off: same maps as the oracle"""
    p = params_with_size(900, **overrides)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    for k in range(4):
        pts, lab, prob, _ = get_scan(k, 900, True)
        hp.processScan(pts, lab, prob, fixed_iterations=6)
        op.process_scan(pts, lab, prob, fixed_iterations=6)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose bits"
        assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k} stats"
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k} surfels"
        frames_equal(hp.frame(2), op.frame(2), f"scan {k} model frame")


def test_pipeline_scans_with_nan_and_inf_points(hip, oracle_lib):
    """sensor drop-outs delivered as NaN / +-inf / zero points are clipped by K1's range and image tests
    (gen_vertexmap.vert:88-93: nothing is written for them) -- identically on both sides"""
    p = params_with_size(900)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    rng = np.random.default_rng(99)
    for k in range(4):
        pts, lab, prob, _ = get_scan(k, 900, True)
        pts = pts.copy()
        bad = rng.choice(pts.shape[0], size=pts.shape[0] // 25, replace=False)
        pts[bad[0::4], 0] = np.nan
        pts[bad[1::4], 1] = np.inf
        pts[bad[2::4], 2] = -np.inf
        pts[bad[3::4], :3] = 0.0
        hp.processScan(pts, lab, prob, fixed_iterations=6)
        op.process_scan(pts, lab, prob, fixed_iterations=6)
        assert np.all(np.isfinite(op.pose()))
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose bits"
        assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k} stats"
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k} surfels"
        for w in (0, 1, 2):
            frames_equal(hp.frame(w), op.frame(w), f"scan {k} frame {w}")


def test_pipeline_convergence_mode(hip, oracle_lib):
    """default.xml stopping tests (no fixed iteration count), no semantics: BASELINE config 1 style"""
    p = params_with_size(900, max_iterations=10)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    for k in range(3):
        pts, lab, prob, _ = get_scan(k, 900, False)
        hp.processScan(pts, lab, prob)
        op.process_scan(pts, lab, prob)
        assert np.array_equal(hp.getCurrentPose(), op.pose())
        assert hp.lastStats().as_dict() == op.last_stats().as_dict()


def test_error_paths(hip):
    p = params_with_size(900)
    ctx = hip.Context(p)
    f = hip.Frame(ctx, 100, 10)
    with pytest.raises(hip.SumaError):
        hip.Preprocessing(ctx).process(np.zeros((4, 4), np.float32), f, None, None, 0)  # wrong frame size
    with pytest.raises(hip.SumaError):
        hip.SurfelMap(ctx).update(np.eye(4), f)
    obj = hip.Frame2Model(ctx)
    with pytest.raises(hip.SumaError):
        obj.jacobianProducts()  # setData not called
    p_bad = params_with_size(1024)
    with pytest.raises(hip.SumaError):
        ctx.set_params(p_bad)  # geometry is fixed at creation


def test_highres_128x4096_pipeline(hip, oracle_lib):
    """BASELINE config 5 geometry (128 beams x 4096 columns): two scans end to end, bit for bit.
    Exercises > 1 pixel per lane in K6, 512 compaction tiles in K10 and H = 128 patches."""
    p = params_with_size(4096, 128, data_fov_down=-25.0, model_fov_down=-25.0)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    for k in range(2):
        pts, lab, prob, _ = get_scan(k, 4096, True, 128)
        hp.processScan(pts, lab, prob, fixed_iterations=5)
        op.process_scan(pts, lab, prob, fixed_iterations=5)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k} pose"
        assert hp.map.size() == op.ctx.map_size() and hp.map.size() > 100000
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k} surfels"
        frames_equal(hp.frame(2), op.frame(2), f"scan {k} model frame")
    assert hp.lastStats().as_dict() == op.last_stats().as_dict()


def test_capacity_overflow_is_reported(hip):
    """the reference silently truncates when transform feedback overflows (SurfelMap.cpp:726-727);
    here the map is truncated the same way but the condition is reported"""
    p = params_with_size(900, max_surfels=5000)
    hp = hip.SurfelMapping(p)
    pts, lab, prob, _ = get_scan(0, 900, True)
    hp.processScan(pts, lab, prob)
    with pytest.raises(hip.SumaError, match="capacity"):
        hp.map.size()


def test_pipeline_model_image_differs_from_data_image(hip, oracle_lib):
    """model_width/height/fov != data_*: separate projections for K4 (model image) and K7 / K9 (data image),
    bilinear taps of K6 across differently sized maps"""
    from semantic_suma_amd.types import default_params
    p = default_params(data_width=900, data_height=64, model_width=1024, model_height=48, model_fov_up=2.0,
                       model_fov_down=-24.0, model_max_depth=70.0)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    for k in range(4):
        pts, lab, prob, _ = get_scan(k, 900, True)
        hp.processScan(pts, lab, prob, fixed_iterations=8)
        op.process_scan(pts, lab, prob, fixed_iterations=8)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k} pose"
        assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k} stats"
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k} surfels"
        frames_equal(hp.frame(2), op.frame(2), f"scan {k} model frame")


def test_model_image_of_two_million_pixels(hip, oracle_lib):
    """a model image of 2048 x 1024 = 2^21 pixels: k_render splits a pixel test's index into row and column with a
    reciprocal while the index stays below 2^21 and with the integer division from there on (its BIG instantiation,
    launch_map_render / run_render) -- the rendered frames, poses and map equal the oracle's either way"""
    from semantic_suma_amd.types import default_params
    p = default_params(data_width=900, data_height=64, model_width=2048, model_height=1024, model_fov_up=3.0,
                       model_fov_down=-25.0)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p, threads=THREADS)
    for k in range(3):
        pts, lab, prob, _ = get_scan(k, 900, True)
        hp.processScan(pts, lab, prob, fixed_iterations=6)
        op.process_scan(pts, lab, prob, fixed_iterations=6)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k} pose"
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k} surfels"
        for w in (1, 2):
            frames_equal(hp.frame(w), op.frame(w), f"scan {k} frame {w}")


def test_tiny_and_odd_image_sizes(hip, oracle_lib):
    """16 beams x 180 columns, a width that is not a multiple of any tile size, and two images whose pixel count is
    not a multiple of the 64-lane wave (450 x 16 = 112.5 waves, 333 x 10): the wave reductions of K6 must not lose the
    lanes beyond the image (round-2 advisor finding)"""
    for (w, h) in ((180, 16), (1000, 40), (450, 16), (333, 10)):
        p = params_with_size(w, h)
        hp = hip.SurfelMapping(p)
        op = oracle_lib.OraclePipeline(p)
        for k in range(3):
            pts, lab, prob, _ = get_scan(k, w, True, h)
            hp.processScan(pts, lab, prob, fixed_iterations=6)
            op.process_scan(pts, lab, prob, fixed_iterations=6)
            assert np.array_equal(hp.getCurrentPose(), op.pose()), f"{w}x{h} scan {k} pose"
            assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"{w}x{h} scan {k} stats"
            assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"{w}x{h} scan {k} surfels"
            for f in (0, 2):
                frames_equal(hp.frame(f), op.frame(f), f"{w}x{h} scan {k} frame {f}")


@pytest.mark.parametrize("weight_function", [1, 2])
def test_loop_closure_verification(hip, oracle_lib, weight_function):
    """device side of SurfelMapping::checkLoopClosure (SurfelMapping.cpp:662-757) against the same sequence
    of oracle primitives: render_inactive -> 3 x (minimize, jacobianProducts, [render_composed, jacobianProducts]).
    Frame2Model::iteration_ is reset by setData (:693, :719) only and counts on across the guesses (Frame2Model.cpp:
    117-123, Objective.h:45-48): with the Tukey weight (2), which is off while iteration_ == 0
    (Frame2Model_jacobians.geom:129), a later guess differs from a fresh minimisation from its first step."""
    p = params_with_size(900, max_iterations=8, weight_function=weight_function)
    ctx, ora, hmap = _run_maps(hip, oracle_lib, p, 900, 4)
    # age the map: everything becomes "inactive" (creation stamp < timestamp - 100)
    surf = hmap.getAllSurfels()
    hmap.upload(surf, 160)
    ora.map_upload(surf, 160)
    T0 = get_scan(0, 900, True)[3]
    pose_prior = np.linalg.inv(T0) @ get_scan(2, 900, True)[3]        # an "old" pose near the revisit
    cur_pose = np.linalg.inv(T0) @ get_scan(3, 900, True)[3]
    pts, lab, prob, _ = get_scan(3, 900, True)
    hf = hip.Frame(ctx, 900, 64)
    hip.Preprocessing(ctx).process(pts, hf, lab, prob, 160)
    of = ora.preprocess(pts, lab, prob, 160, ora.frame())
    O = np.linalg.inv(pose_prior) @ cur_pose
    O[2, 3] = 0.0
    Rz = O.copy()
    Rz[:2, :2] = -Rz[:2, :2]                                          # second guess: rotated by 180 degrees
    half = O.copy()
    half[:2, 3] *= 0.5
    inits = [O, Rz, half]
    ct = -1.0
    res = hip.loop_closure_verify(ctx, hf, pose_prior, inits, cur_pose, ct)
    # the same sequence on the oracle
    ora.map_render_inactive(pose_prior.astype(np.float32), ct)
    model = ora.map_frame(0)
    n_passed = 0
    it = 0  # Frame2Model::iteration_
    fresh_differs = False
    for k, init in enumerate(inits):
        T, _, st = ora.minimize(of, model, init, iteration0=it)
        if it > 0 and not np.array_equal(ora.minimize(of, model, init)[0], T):
            fresh_differs = True
        it += st.iterations + (1 if st.converged else 0)
        _, _, _, _, s0 = ora.jacobian_products(of, model, T, it)
        assert np.array_equal(res[k]["gn_pose"], T), f"guess {k} pose"
        a = res[k]["after_minimize"]
        assert (a["valid"], a["outlier"], a["inlier"], a["invalid"], a["error"]) == (s0.valid, s0.outlier, s0.inlier, s0.invalid, s0.error)
        valid_ratio = np.float32(s0.valid) / np.float32(s0.valid + s0.invalid)
        outlier_ratio = np.float32(s0.outlier) / np.float32(s0.outlier + s0.inlier)
        passed = bool(valid_ratio > np.float32(0.2) and outlier_ratio < np.float32(0.85))
        assert res[k]["passed"] == passed
        pose_old = (pose_prior @ T).astype(np.float32)
        assert np.array_equal(res[k]["pose_old"], pose_old)
        if passed:
            n_passed += 1
            ora.map_render_composed(pose_old, cur_pose.astype(np.float32), ct)
            model = ora.map_frame(2)   # the reference leaves the objective on the composed frame
            it = 0                     # ... through setData, which resets iteration_
            Fc, _, JtJ, _, sc = ora.jacobian_products(of, model, np.eye(4), 0)
            c = res[k]["composed"]
            assert (c["valid"], c["outlier"], c["invalid"], c["error"]) == (sc.valid, sc.outlier, sc.invalid, sc.error)
            assert np.array_equal(res[k]["JtJ"], JtJ)
    assert n_passed >= 1, "test setup: at least one guess should pass the gates"
    if weight_function == 2 and n_passed < len(inits):
        assert fresh_differs, "test setup: the running iteration counter should matter to a Tukey-weighted later guess"
    assert fresh_differs == (weight_function == 2 and fresh_differs)  # Huber never reads the counter


def test_pose_table_download(hip, oracle_lib):
    """suma_map_download_poses = SurfelMap::poses_ (SurfelMap.h:205-208, written at SurfelMap.cpp:494-495): entry t is the
    pose scan t was integrated with, as float -- the trajectory bench.py feeds to the KITTI devkit metric"""
    p = params_with_size(900)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p)
    assert hp.map.poses().shape == (0, 4, 4)
    want = []
    for k in range(5):
        pts, lab, prob, _ = get_scan(k, 900, True)
        hp.processScan(pts, lab, prob, fixed_iterations=6)
        op.process_scan(pts, lab, prob, fixed_iterations=6)
        want.append(hp.getCurrentPose().astype(np.float32))
    got = hp.map.poses()
    assert got.shape == (5, 4, 4) and got.dtype == np.float32
    assert_bit_equal(got, np.stack(want), "pose table")
    ora = op.ctx.map_poses(5).reshape(-1, 4, 4).transpose(0, 2, 1)
    assert_bit_equal(got, ora, "pose table vs oracle")


def _same_loop_results(a, b, what):
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        for key in ("gn_pose", "pose_old", "JtJ"):
            assert x[key].tobytes() == y[key].tobytes(), f"{what}: guess {k} {key}"
        for key in ("after_minimize", "composed", "passed"):
            assert x[key] == y[key], f"{what}: guess {k} {key}: {x[key]} vs {y[key]}"


@pytest.mark.parametrize("max_iterations,weight_function", [(8, 1), (0, 1), (8, 2)])
def test_batched_loop_closure_verification_equals_the_sequential_form(hip, oracle_lib, max_iterations, weight_function):
    """suma_loop_closure_verify runs the initial guesses of SurfelMapping.cpp:679-757 as ONE batched
    Gauss-Newton chain, speculatively against oldMapFrame(), and redoes the guesses behind the first one that passes
    against composedFrame() (the reference's setData at :718-719).  Whatever the order of the guesses and wherever the
    gates fall -- nobody passes (one round), everybody passes (a round per guess), the first / the last passes, more
    guesses than one round would need -- the results are those of the literal one-by-one sequencing
    (suma_loop_closure_verify_serial, which test_loop_closure_verification's oracle sequence pins), bit for bit.
    max_iterations = 0 is the reference's "until convergence" mode (chunked launches, LieGaussNewton.cpp:27)."""
    p = params_with_size(900, max_iterations=max_iterations, weight_function=weight_function)
    ctx, ora, hmap = _run_maps(hip, oracle_lib, p, 900, 4)
    surf = hmap.getAllSurfels()
    hmap.upload(surf, 160)
    T0 = get_scan(0, 900, True)[3]
    pose_prior = np.linalg.inv(T0) @ get_scan(2, 900, True)[3]
    cur_pose = np.linalg.inv(T0) @ get_scan(3, 900, True)[3]
    pts, lab, prob, _ = get_scan(3, 900, True)
    hf = hip.Frame(ctx, 900, 64)
    hip.Preprocessing(ctx).process(pts, hf, lab, prob, 160)
    O = np.linalg.inv(pose_prior) @ cur_pose
    O[2, 3] = 0.0
    Rz = O.copy()
    Rz[:2, :2] = -Rz[:2, :2]
    half = O.copy()
    half[:2, 3] *= 0.5
    far = O.copy()
    far[:2, 3] += 30.0
    ct = -1.0
    ref_gates, nobody, everybody = (0.2, 0.85), (2.0, 0.85), (-1.0, 2.0)
    if max_iterations:
        g = [O, Rz, half, far]
        cases = [(order, gates) for order in ([0, 1, 2], [1, 2, 0], [3, 1, 0], [3, 3, 0], [0], [1, 3, 2, 0, 2, 1, 0])
                 for gates in (ref_gates, nobody, everybody)]
        # more guesses than one batched launch holds (SUMA_MAX_HYP = 64): the rounds continue behind the first 64, with
        # the iteration counter that the chains in front have moved
        cases += [([3, 1, 2] * 23 + [0], nobody), ([3, 1] * 33 + [0, 2, 0], ref_gates)]
    else:
        # "until convergence" has no iteration cap, and on this four-scan map many chains end in a limit cycle (300
        # iterations without meeting LieGaussNewton.cpp:64-66 -- the reference would iterate for ever as well): only
        # guesses whose chains were seen to converge, against oldMapFrame() and against the composed frame of a first pass
        def scaled(f):
            q = O.copy()
            q[:2, 3] *= f
            return q
        g = [half, scaled(0.45), scaled(0.4), scaled(0.6)]
        cases = [([0, 1], ref_gates), ([0, 3], ref_gates), ([0], ref_gates), ([2, 1, 0], nobody), ([1, 0, 2, 1, 0], nobody)]
    n_rounds_seen = set()
    for order, gates in cases:
        inits = [g[i] for i in order]
        a = hip.loop_closure_verify(ctx, hf, pose_prior, inits, cur_pose, ct, *gates)
        b = hip.loop_closure_verify(ctx, hf, pose_prior, inits, cur_pose, ct, *gates, serial=True)
        _same_loop_results(a, b, f"order {order} gates {gates}")
        n_rounds_seen.add((len(order), sum(r["passed"] for r in a)))
    assert any(n > 1 and k == 0 for n, k in n_rounds_seen) and any(n > 1 and k >= 1 for n, k in n_rounds_seen)
    if max_iterations:
        assert any(n > 1 and k == n for n, k in n_rounds_seen)


def test_history_belongs_to_the_optimizer_that_minimised_last(hip, oracle_lib):
    """LieGaussNewton::history() is fetched lazily from ONE device buffer per context; the reference keeps history_ per
    optimizer object (LieGaussNewton.h:72).  An object whose chain has been overwritten by another minimisation must say
    so instead of handing out the other chain's poses (round-4 advisor); read in time, it holds the oracle's history."""
    p = params_with_size(900)
    ora, f0, f1 = _model_and_data(oracle_lib, p, 900, True)
    ctx = hip.Context(p)
    h0, h1 = (hip.Frame(ctx, 900, 64) for _ in range(2))
    for h, f in ((h0, f0), (h1, f1)):
        h.set(f.vertex, f.normal, f.semantic)
    obj = hip.Frame2Model(ctx)
    obj.setData(h1, h0)
    a, b = hip.LieGaussNewton(ctx), hip.LieGaussNewton(ctx)
    T0 = np.eye(4)
    T0[0, 3] = 1.0
    a.minimize(obj, T0)
    _, hist, _ = ora.minimize(f1, f0, T0)
    assert np.array_equal(a.history(), hist)  # fetched before anything else ran: the oracle's chain, pose by pose
    a.minimize(obj, T0)
    b.minimize(obj, T0 @ T0)
    with pytest.raises(RuntimeError, match="overwritten"):
        a.history()
    assert b.history().shape[0] == ora.minimize(f1, f0, T0 @ T0)[1].shape[0]


def test_new_frame_behind_a_backlog_of_the_ctx_stream(hip, oracle_lib):
    """A frame created while the ctx stream still holds queued work: its zeroing memset is ctx-stream work, and the
    side-stream preprocessing into it has to be ordered behind it (round-4 advisor: last_access stayed 0 and the memset
    could land on the freshly written maps).  The maps must be the oracle's whatever the backlog."""
    p = params_with_size(2048)
    pipe = hip.SurfelMapping(p)
    ctx = pipe.ctx
    for k in range(6):  # a map worth ~0.4 ms of queued kernels per scan
        pipe.processScan(*get_scan(k, 2048, True)[:3], fixed_iterations=10)
    ora = oracle_lib.Oracle(p)
    pts, lab, prob, _ = get_scan(7, 2048, True)
    want = ora.preprocess(pts, lab, prob, 30, ora.frame())
    pre = hip.Preprocessing(ctx)
    for rep in range(8):
        for k in range(3):  # backlog: three scans enqueued, none waited for
            pipe.processScan(*get_scan(6 + k, 2048, True)[:3], fixed_iterations=10)
        fresh = hip.Frame(ctx, 2048, 64)  # memset queued behind the backlog
        pre.process(pts, fresh, lab, prob, 30)
        assert_bit_equal(fresh.vertex, want.vertex, f"vertex map, repetition {rep}")
        assert_bit_equal(fresh.normal, want.normal, f"normal map, repetition {rep}")


def test_two_objectives_with_their_own_gates(hip, oracle_lib):
    """objective_ and recovery_ = Frame2Model(fallback_params) (SurfelMapping.cpp:87-94) live side by side on one
    context: each object sends its own gates and its own frame pair before a launch (round-1 advisor finding)."""
    p = params_with_size(900)
    p_fb = params_with_size(900, icp_max_distance=p.fallback_max_distance, icp_max_angle=p.fallback_max_angle)
    ora, f0, f1 = _model_and_data(oracle_lib, p, 900, True)
    ora_fb = oracle_lib.Oracle(p_fb)
    pts2, l2, pr2, _ = get_scan(3, 900, True)
    f2 = ora.preprocess(pts2, l2, pr2, 22, ora.frame())
    ctx = hip.Context(p)
    h0, h1, h2 = (hip.Frame(ctx, 900, 64) for _ in range(3))
    for h, f in ((h0, f0), (h1, f1), (h2, f2)):
        h.set(f.vertex, f.normal, f.semantic)
    objective, recovery = hip.Frame2Model(ctx), hip.Frame2Model(ctx, p_fb)
    objective.setData(h1, h0)
    recovery.setData(h2, h1)  # another pair: the last setData on the context must not leak into `objective`
    gn = hip.LieGaussNewton(ctx)
    T0 = np.eye(4)
    T0[0, 3] = 1.0
    for _ in range(2):  # interleaved
        gn.minimize(objective, T0)
        To, _, st = ora.minimize(f1, f0, T0)
        assert np.array_equal(gn.pose(), To) and objective.valid() == st.valid and objective.outlier() == st.outlier
        info = gn.information()
        assert np.array_equal(info, info.T) and np.all(np.diag(info) > 0)
        gn.minimize(recovery, T0 @ T0)
        To, _, st = ora_fb.minimize(f2, f1, T0 @ T0)
        assert np.array_equal(gn.pose(), To) and recovery.outlier() == st.outlier
    # information() = JtJ of the last step: jacobianProducts at the pose before the last increment reproduces it
    p1 = params_with_size(900, max_iterations=1)
    ctx.set_params(p1)
    gn.minimize(objective, T0)
    info = gn.information()
    objective.initialize(T0)
    _, JtJ, _ = objective.jacobianProducts()
    assert np.array_equal(info, JtJ)
    # setParameter reaches only its own object
    recovery.setParameter("icp-max-distance", 0.05)
    recovery.initialize(T0 @ T0)
    recovery.jacobianProducts()
    p_tight = params_with_size(900, icp_max_distance=0.05, icp_max_angle=p.fallback_max_angle)
    _, acc, _, _, _ = oracle_lib.Oracle(p_tight).jacobian_products(f2, f1, T0 @ T0, 0)
    assert np.array_equal(recovery.acc, acc)
    objective.initialize(T0)
    objective.jacobianProducts()
    assert np.array_equal(objective.acc, ora.jacobian_products(f1, f0, T0, 0)[1])
    assert hip.LieGaussNewton.reason(-1).startswith("Maximum") and hip.LieGaussNewton.reason(0) == "no error"


def test_minimize_until_convergence(hip, oracle_lib):
    """max iterations = 0 means "until convergence" (LieGaussNewton.cpp:27): launches are enqueued in chunks of 32 and
    the done flags polled -- no silent cap (round-1 advisor finding)"""
    p0 = params_with_size(900)
    ora, f0, f1 = _model_and_data(oracle_lib, p0, 900, False)
    ctx = hip.Context(p0)
    h0, h1 = hip.Frame(ctx, 900, 64), hip.Frame(ctx, 900, 64)
    h0.set(f0.vertex, f0.normal, f0.semantic)
    h1.set(f1.vertex, f1.normal, f1.semantic)
    obj = hip.Frame2Model(ctx)
    obj.setData(h1, h0)
    gn = hip.LieGaussNewton(ctx)
    T0 = np.eye(4)
    T0[0, 3] = 1.0
    longest = 0
    for eps, delta in ((0.5, 2e-3), (0.05, 1e-3)):  # the oracle converges after 27 / 136 iterations
        # (the oracle has no cap either: ask it with a finite budget first)
        ora.set_params(params_with_size(900, max_iterations=400, stopping_threshold=eps, delta=delta))
        To, _, st = ora.minimize(f1, f0, T0)
        assert st.converged == 1, "test setup"
        ctx.set_params(params_with_size(900, max_iterations=0, stopping_threshold=eps, delta=delta))
        gn.minimize(obj, T0, history_cap=0)
        assert np.array_equal(gn.pose(), To), f"thresholds {eps} / {delta}"
        assert gn.stats.converged == 1 and gn.iterationCount() == st.iterations
        longest = max(longest, st.iterations)
    assert longest > 64, "no chain crossed a chunk boundary"


def test_frame_swap_and_viewer_exports(hip, oracle_lib):
    """suma_frame_swap, suma_frame_export, suma_map_export_surfels / _data_surfels: the buffers a viewer would
    register with hipGraphicsGLRegisterBuffer / Image hold exactly what the download entry points return."""
    width = 900
    p = params_with_size(width)
    hp = hip.SurfelMapping(p)
    for k in range(3):
        pts, lab, prob, _ = get_scan(k, width, True)
        hp.processScan(pts, lab, prob, fixed_iterations=5)
    ctx = hp.ctx
    surfels = hp.map.getAllSurfels()
    d_ptr, n = hp.map.getModelSurfels()
    assert n == surfels.shape[0] and d_ptr
    assert ctx.device_download(d_ptr, 64 * n).tobytes() == surfels.tobytes()
    d_ptr2, first, n_data = hp.map.getDataSurfels()
    su, sn, _, _ = hp.map.counts()
    assert d_ptr2 == d_ptr and first + n_data == n and 0 < n_data <= sn
    tail = surfels[first:]
    assert np.all(tail["timestamp"] == 2) and np.all(tail["count"] == 2.0)  # created by the last scan
    f = hp.frame(2)
    for which in range(3):
        ptr, w, h, rb = f.export(which)
        assert (w, h, rb) == (width, 64, 16 * width)
        assert ctx.device_download(ptr, rb * h).tobytes() == f.download(which).tobytes()
    a, b = hip.Frame(ctx, width, 64), hip.Frame(ctx, width, 64)
    va = np.random.default_rng(0).random((64, width, 4), dtype=np.float32)
    a.upload(0, va)
    b.upload(0, 2 * va)
    a.swap(b)
    assert np.array_equal(a.download(0), 2 * va) and np.array_equal(b.download(0), va)
    with pytest.raises(hip.SumaError):
        a.swap(hip.Frame(ctx, 100, 10))


def test_async_ingest_equals_resident_scans(hip):
    """suma_pipeline_prefetch_scan / process_prefetched (pinned double buffer + copy stream + ingest thread): same
    poses and map as the blocking host path, with scan k+1 staged while scan k runs (KITTIReader.cpp:136-203)."""
    width, n = 900, 9
    p = params_with_size(width)
    scans_ = [get_scan(k, width, True)[:3] for k in range(n)]
    ref = hip.SurfelMapping(p)
    for pts, lab, prob in scans_:
        ref.processScan(pts, lab, prob, fixed_iterations=10)
    a = hip.SurfelMapping(p)
    seen = []
    assert a.processSequence(scans_, fixed_iterations=10, on_scan=lambda k, s: seen.append(s.getCurrentPose().copy())) == n
    assert np.array_equal(a.getCurrentPose(), ref.getCurrentPose()) and len(seen) == n
    assert a.map.getAllSurfels().tobytes() == ref.map.getAllSurfels().tobytes()
    # one-call form, scans of different sizes (staging grows), no labels, an empty scan
    b, c = hip.SurfelMapping(p), hip.SurfelMapping(p)
    odd = [(scans_[0][0][:5000], None, None), (scans_[1][0], None, None), (np.zeros((0, 4), np.float32), None, None),
           (scans_[2][0], None, None)]
    for pts, lab, prob in odd:
        b.ctx.check(b.L.suma_pipeline_process_scan_async(b.h, pts.ctypes.data if pts.size else None, None, None,
                                                         pts.shape[0], 5), "async")
        c.processScan(pts, lab, prob, fixed_iterations=5)
    assert b.getCurrentPose().tobytes() == c.getCurrentPose().tobytes()  # NaN after the empty scan, on both paths
    assert b.map.getAllSurfels().tobytes() == c.map.getAllSurfels().tobytes()
    # misuse is reported
    with pytest.raises(hip.SumaError):
        a.processPrefetched()
    a.prefetchScan(*scans_[0])
    a.prefetchScan(*scans_[1])
    a.prefetchScan(*scans_[2])
    with pytest.raises(hip.SumaError):
        a.prefetchScan(*scans_[3])  # all three slots staged


def test_native_scan_loop_equals_scan_by_scan_calls(hip):
    """suma_pipeline_run_scans (include/suma_runner.h): the caller's loop over processScan in native code, on an existing
    pipeline, for resident and for host scans -- the same poses and the same map bytes as one call per scan, the count
    of scans done, per-call host times filled in."""
    width, n = 900, 8
    p = params_with_size(width)
    scans_ = [get_scan(k, width, True)[:3] for k in range(n)]
    ref = hip.SurfelMapping(p)
    for sc in scans_:
        ref.processScan(*sc, fixed_iterations=10)
    a, b = hip.SurfelMapping(p), hip.SurfelMapping(p)
    dev = [tuple(a.ctx.device_array(x) for x in sc) + (sc[0].shape[0],) for sc in scans_]
    a.processScanDevice(*dev[0], fixed_iterations=10)  # the loop continues a sequence that has already begun
    assert a.runScans(dev[1:], True, fixed_iterations=10) == n - 1
    secs = np.zeros(n, dtype=np.float64)
    assert b.runScans(scans_, False, fixed_iterations=10, call_seconds=secs) == n
    assert np.all(secs > 0.0) and secs.sum() < 5.0
    for q in (a, b):
        assert np.array_equal(q.getCurrentPose(), ref.getCurrentPose())
        assert q.map.getAllSurfels().tobytes() == ref.map.getAllSurfels().tobytes()
    assert a.runScans([], True) == 0


_GATHER_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from semantic_suma_amd import core
from semantic_suma_amd.types import params_with_size
D = C.CDLL(os.path.join(os.path.dirname(core.LIB_PATH), "libsuma_hip_dist.so"))
vp = C.c_void_p
D.suma_dist_unique_id.argtypes = [vp]
D.suma_dist_comm_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
D.suma_dist_comm_destroy.argtypes = [vp]
D.suma_gather_poses.argtypes = [vp, vp, vp, vp]
D.suma_gather.argtypes = [vp, vp, vp, C.c_uint32, vp]
D.suma_dist_last_error.restype = C.c_char_p
D.suma_dist_last_error.argtypes = [vp]
ctx = core.Context(params_with_size(900))
uid = C.create_string_buffer(128)
assert D.suma_dist_unique_id(uid) == 0, D.suma_dist_last_error(None)
comm = vp()
assert D.suma_dist_comm_create(uid, 1, 0, C.byref(comm)) == 0, D.suma_dist_last_error(None)
pose = np.arange(16, dtype=np.float64) + 0.5
out = np.zeros(16, dtype=np.float64)
assert D.suma_gather_poses(ctx.h, comm, pose.ctypes.data, out.ctypes.data) == 0, D.suma_dist_last_error(comm)
assert np.array_equal(out, pose)
payload = np.linspace(0, 1, 21)
out2 = np.zeros(21)
assert D.suma_gather(ctx.h, comm, payload.ctypes.data, 21, out2.ctypes.data) == 0 and np.array_equal(out2, payload)
assert D.suma_gather(ctx.h, comm, payload.ctypes.data, 65, out2.ctypes.data) != 0  # over the 64-double limit
# the exchange step of the native hypothesis runner (suma_run_hypotheses) over RCCL: a table of any size, summed over
# the ranks (world 1: returned as is), directly and through the suma_exchange_fn signature
D.suma_dist_allreduce_sum.argtypes = [vp, vp, C.c_uint32, vp]
D.suma_dist_exchange.argtypes = [vp, vp, vp, C.c_uint32]
table = np.random.default_rng(3).normal(size=8 * 18)
got = np.zeros_like(table)
assert D.suma_dist_allreduce_sum(comm, table.ctypes.data, table.size, got.ctypes.data) == 0, D.suma_dist_last_error(comm)
assert np.array_equal(got, table)
got[:] = 0
assert D.suma_dist_exchange(comm, table.ctypes.data, got.ctypes.data, table.size) == 0 and np.array_equal(got, table)
# and the runner driven with it from this process (one rank: the callback is not needed, but must be accepted)
from semantic_suma_amd import synth
from semantic_suma_amd.distributed import hypothesis_perturbations, run_hypotheses_hip
p9 = params_with_size(900, max_iterations=6)
scans = [synth.generate_scan(k, n_azimuth=900)[:3] for k in range(3)]
poses, winners = run_hypotheses_hip(p9, scans, 4)
assert winners[0] == -1 and all(0 <= w < 4 for w in winners[1:]) and 1.5 < poses[-1][0, 3] < 3.0
D.suma_dist_comm_destroy(comm)
print("gather ok")
"""


def test_gather_poses_rccl_world_1():
    """libsuma_hip_dist.so: the C-ABI gather (RCCL all-gather on the ctx stream).  A 1-GPU box can only host a
    communicator of size 1 (RCCL refuses two ranks on one device); the N-rank path is exercised by bench.py --gpus N.
    Runs in a fresh interpreter, like the C++ host it is meant for: a process that has already imported torch carries
    torch's bundled ROCm runtime next to the system one, and RCCL must initialise against the runtime the HIP
    library uses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _GATHER_SCRIPT, root], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "gather ok" in out.stdout, out.stderr[-2000:]
