"""GPU parity of the phase API of the scan pipeline (suma_pipeline_begin_scan / _update_pose / _update_map and the
loop-closure hooks between them) -- what SurfelMapping::processScan needs when config/default.xml's close-loops = true
(SurfelMapping.cpp:175-204, 211-250, 546-574, 662-757).  Everything through the C-ABI, bit for bit against the oracle."""
import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import params_with_size

import loop_scenario as ls

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from semantic_suma_amd import core
    core.lib()
    return core


def test_phases_are_the_single_call(hip):
    """process_scan IS begin + update_pose + update_map: same bits from host pointers, device pointers and staged scans"""
    p = params_with_size(900)
    a, b, c, d = (hip.SurfelMapping(p) for _ in range(4))
    scans = [get_scan(k, 900, True)[:3] for k in range(5)]
    for k, (pts, lab, prob) in enumerate(scans):
        a.processScan(pts, lab, prob, fixed_iterations=8)
        b.beginScan(pts, lab, prob)
        b.updatePose(8)
        b.updateMap()
        dev = [c.ctx.device_array(x) for x in (pts, lab, prob)]
        c.beginScanDevice(dev[0], dev[1], dev[2], pts.shape[0])
        c.updatePose(8)
        c.updateMap()
        d.prefetchScan(pts, lab, prob)
        d.beginPrefetched()
        d.updatePose(8)
        d.updateMap()
        for o in (b, c, d):
            assert np.array_equal(a.getCurrentPose(), o.getCurrentPose()), f"scan {k} pose"
            assert a.lastStats().as_dict() == o.lastStats().as_dict(), f"scan {k} stats"
            assert a.map.getAllSurfels().tobytes() == o.map.getAllSurfels().tobytes(), f"scan {k} surfels"
    assert np.array_equal(b.getPose(3), b.getPose(4))  # lastPose_old_ == lastPose_ without closures
    with pytest.raises(hip.SumaError):
        b.updateMap()                                   # phase order is checked
    with pytest.raises(hip.SumaError):
        b.trackLoopClosure()
    b.beginScan(*scans[0])
    with pytest.raises(hip.SumaError):
        b.beginScan(*scans[0])


def test_se3_log_host_math(hip, oracle_lib):
    rng = np.random.default_rng(11)
    for k in range(100):
        x = rng.uniform(-1, 1, 6)
        T = oracle_lib.se3_exp(x)
        assert np.array_equal(hip.se3_log(T), oracle_lib.se3_log(T))


def _compare_loop(a, b, what):
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x["gn_pose"], y["gn_pose"], equal_nan=True), f"{what} guess {k} pose"
        assert x["after_minimize"] == y["after_minimize"], f"{what} guess {k} stats"
        assert x["passed"] == y["passed"]
        assert np.array_equal(x["pose_old"], y["pose_old"], equal_nan=True)
        assert x["composed"] == y["composed"], f"{what} guess {k} composed stats"
        assert np.array_equal(x["JtJ"], y["JtJ"]), f"{what} guess {k} information"


def _compare_track(x, y, what):
    assert np.array_equal(x["increment_old"], y["increment_old"]), f"{what} increment"
    assert x["after_minimize"] == y["after_minimize"], f"{what} stats"
    assert x["increment_difference"] == y["increment_difference"] and x["passed"] == y["passed"]
    assert np.array_equal(x["pose_old"], y["pose_old"])
    assert x["composed"] == y["composed"] and np.array_equal(x["JtJ"], y["JtJ"]), f"{what} composed"


def test_scripted_loop_closing_run(hip, oracle_lib):
    """123 scans on a closed circle with the hooks of tests/loop_scenario.py: a candidate verified from three initial
    guesses, four scans of tracking, an optimised trajectory integrated.  Pose bits, statistics and the whole surfel
    buffer after every scan; the model frames around the hooks (currentPose_old_ != currentPose_new_ renders both
    parts of the map)."""
    W, H = 900, 64
    p = params_with_size(W, H)
    lap = ls.lap_scans()
    k_detect = lap + 15
    n = k_detect + 11
    hp, op = hip.SurfelMapping(p), oracle_lib.OraclePipeline(p, threads=16)
    pipes = [ls.HipPipe(hp), ls.OraclePipe(op)]
    real_verify, real_track = ls.HipPipe.verify, ls.HipPipe.track
    seen = dict(verify=None, tracks=[])

    def on_scan(k, pp):
        h, o = pp
        assert np.array_equal(h.pose(0), o.pose(0)), f"scan {k}: pose bits"
        for w in (1, 2, 3):
            assert np.array_equal(h.pose(w), o.pose(w)), f"scan {k}: pose {w}"
        assert h.stats() == o.stats(), f"scan {k}: statistics"
        if k % 8 == 0 or k >= k_detect - 1:
            assert h.surfels().tobytes() == o.surfels().tobytes(), f"scan {k}: surfels"
            for w in (1, 2):
                for m, (x, y) in enumerate(zip(h.frame(w), o.frame(w))):
                    assert x.tobytes() == y.tobytes(), f"scan {k}: frame {w} map {m}"

    # capture both sides' hook results for the bitwise comparison
    res_h, res_o = [], []
    ls.HipPipe.verify = lambda self, prior, inits: res_h.append(("v", real_verify(self, prior, inits))) or res_h[-1][1]
    ls.HipPipe.track = lambda self: res_h.append(("t", real_track(self))) or res_h[-1][1]
    ov, ot = ls.OraclePipe.verify, ls.OraclePipe.track
    ls.OraclePipe.verify = lambda self, prior, inits: res_o.append(("v", ov(self, prior, inits))) or res_o[-1][1]
    ls.OraclePipe.track = lambda self: res_o.append(("t", ot(self))) or res_o[-1][1]
    try:
        log = ls.run(pipes, W, H, n, k_detect, 4, k_detect + 8, iterations=8, on_scan=on_scan)
    finally:
        ls.HipPipe.verify, ls.HipPipe.track = real_verify, real_track
        ls.OraclePipe.verify, ls.OraclePipe.track = ov, ot
    assert len(res_h) == len(res_o) == 5
    for (kh, rh), (ko, ro) in zip(res_h, res_o):
        assert kh == ko
        if kh == "v":
            _compare_loop(rh, ro, "verify")
        else:
            _compare_track(rh, ro, "track")
    assert any(g["passed"] for g in log["verify"]) and any(t["passed"] for t in log["tracks"])
    assert log["integrated"] and log["moved_pose_old"] >= 2
    assert hp.map.size() == op.ctx.map_size() and hp.map.size() > 100000
    gt = np.linalg.inv(ls.circle_pose(0)) @ ls.circle_pose(n - 1)
    assert np.linalg.norm((np.linalg.inv(hp.getCurrentPose()) @ gt)[:3, 3]) < 1.0


def test_pipeline_reset_is_a_fresh_pipeline(hip):
    """SurfelMapping::reset (SurfelMapping.cpp:131-169): after a reset -- also one issued between update_pose and update_map,
    with an index-map splat pending -- the pipeline reproduces a new object's results bit for bit"""
    p = params_with_size(900)
    scans = [get_scan(k, 900, True)[:3] for k in range(4)]
    fresh = hip.SurfelMapping(p)
    for sc in scans:
        fresh.processScan(*sc, fixed_iterations=8)
    used = hip.SurfelMapping(p)
    for sc in scans[::-1]:
        used.processScan(*sc, fixed_iterations=8)
    used.beginScan(*scans[0])
    used.updatePose(8)      # leaves the fused K7 splat in the z-buffer
    used.reset()
    assert used.timestamp() == 0 and used.map.size() == 0 and np.array_equal(used.getCurrentPose(), np.eye(4))
    for sc in scans:
        used.processScan(*sc, fixed_iterations=8)
    assert np.array_equal(used.getCurrentPose(), fresh.getCurrentPose())
    assert used.lastStats().as_dict() == fresh.lastStats().as_dict()
    assert used.map.getAllSurfels().tobytes() == fresh.map.getAllSurfels().tobytes()


def test_cache_arena_is_compacted_when_it_runs_full(hip, oracle_lib):
    """The submap cache is a bump allocator in HBM; a tile that is extracted again leaves its old block behind (the
    reference overwrites a std::vector, SurfelMap.cpp:733-734).  Small tiles and a small arena on the closed circle:
    tiles are extracted, re-appended and extracted again lap after lap, the arena runs full, the live tiles are moved
    into a fresh one -- and the map stays the oracle's, bit for bit."""
    W, H = 900, 64
    kw = dict(submap_extent=4.0, submap_dimension=2, cache_surfels=900_000)
    p = params_with_size(W, H, **kw)
    hp, op = hip.SurfelMapping(p), oracle_lib.OraclePipeline(p, threads=16)
    n = 2 * ls.lap_scans() + 20
    for k in range(n):
        sc = ls.scan(k, W, H)
        hp.processScan(*sc, fixed_iterations=6)
        op.process_scan(*sc, fixed_iterations=6)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose"
        if k % 10 == 9 or k == n - 1:
            su, sn, cached, origin = hp.map.counts()
            assert cached == op.ctx.map_cached_surfels() and origin == op.ctx.map_submap_origin(), f"scan {k}: cache"
            assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k}: surfels"
    used, cap, compactions = hp.map.cache_stats()
    assert cap == 900_000 and compactions >= 1, (used, cap, compactions)
    assert used <= cap


@pytest.mark.parametrize("fused", [True, False])
def test_parked_tiles_are_the_oracles(hip, oracle_lib, fused, monkeypatch):
    """extractSurfels (SurfelMap.cpp:708-742): every tile that leaves the window is parked with exactly the records, in
    exactly the order, of the reference's transform feedback.  Usually the extraction rides on the update's own
    stream-out (k9_update<true> / k10_generate<true>); when a parked tile comes back in the same scan -- second lap of the
    circle -- it falls back to the K12 launch.  Both routes, every parked tile compared after every extraction."""
    if fused:
        monkeypatch.delenv("SUMA_NO_FUSED_EXTRACT", raising=False)
    else:
        monkeypatch.setenv("SUMA_NO_FUSED_EXTRACT", "1")
    W, H = 900, 64
    p = params_with_size(W, H, submap_extent=4.0, submap_dimension=2)
    hp, op = hip.SurfelMapping(p), oracle_lib.OraclePipeline(p, threads=16)
    n = ls.lap_scans() + 40
    seen, extractions, compared = set(), 0, 0
    for k in range(n):
        sc = ls.scan(k, W, H)
        hp.processScan(*sc, fixed_iterations=6)
        op.process_scan(*sc, fixed_iterations=6)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose"
        count, (ei, ej) = op.ctx.map_last_extraction()
        if count != extractions:
            extractions = count
            seen.add((int(ei), int(ej)))
            want = op.ctx.map_cache_tile(int(ei), int(ej))
            got = hp.map.cached_tile(int(ei), int(ej))
            assert got.shape[0] == want.shape[0], f"scan {k}: tile ({ei},{ej}) holds {got.shape[0]} surfels, oracle {want.shape[0]}"
            assert got.tobytes() == want.tobytes(), f"scan {k}: parked tile ({ei},{ej}) differs"
            compared += want.shape[0]
    assert extractions > 20 and compared > 50_000, (extractions, compared)
    for (i, j) in seen:  # the tiles as they stand at the end (re-extracted ones hold their latest block)
        assert hp.map.cached_tile(i, j).tobytes() == op.ctx.map_cache_tile(i, j).tobytes(), f"tile ({i},{j}) at the end"
    assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes()
    assert hp.map.cached_tile(1000, 1000).shape[0] == 0
