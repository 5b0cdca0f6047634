"""GPU parity at the sizes the bench and BASELINE config 5 run at (run with -m gpu on an MI355X).

* test_long_sequence_parity    -- the bench workload itself (64x2048, semantic ICP, 10 GN iterations) for 200 scans
                                  against the 16-thread oracle: pose bits and statistics every scan, the whole
                                  surfel buffer every 10 scans; crosses >= 3 submap-origin shifts with tile
                                  extraction / re-appending and reaches the ~1 M-surfel steady state.
* test_index_above_2_24        -- 20 M and 50 M surfels (BASELINE configs[4] verbatim) at 128x4096: one update + one render against the oracle; the
                                  uint32 index map must name surfels beyond 2^24 (a float index map, as in
                                  gen_indexmap.vert:79, could not).
* test_two_pipelines_one_gpu   -- two pipelines on two host threads sharing one device, interleaved, each
                                  bit-equal to its solo run (per-context pinned records, tickets, epochs).
"""
import os
import threading

import numpy as np
import pytest

from conftest import assert_bit_equal, get_scan
from semantic_suma_amd.types import SURFEL_DTYPE, params_with_size

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def hip():
    from semantic_suma_amd import core
    core.lib()
    return core


def test_long_sequence_parity(hip, oracle_lib):
    from semantic_suma_amd import synth
    W, N = 2048, 200
    p = params_with_size(W)
    hp = hip.SurfelMapping(p)
    op = oracle_lib.OraclePipeline(p, threads=THREADS)
    origins, max_map, cached_seen = set(), 0, 0
    for k in range(N):
        pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W)
        hp.processScan(pts, lab, prob, fixed_iterations=10)
        op.process_scan(pts, lab, prob, fixed_iterations=10)
        assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose bits"
        assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k}: statistics"
        su, sn, cached, origin = hp.map.counts()
        assert (su, sn) == op.ctx.map_counts() and cached == op.ctx.map_cached_surfels(), f"scan {k}: counts"
        assert origin == op.ctx.map_submap_origin(), f"scan {k}: submap origin"
        origins.add(origin)
        cached_seen = max(cached_seen, cached)
        max_map = max(max_map, hp.map.size())
        if k % 10 == 9 or k == N - 1:
            assert hp.map.size() == op.ctx.map_size(), f"scan {k}: map size"
            assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k}: surfel bytes"
    for w in (0, 1, 2):
        for m, name in enumerate(("vertex", "normal", "semantic")):
            assert_bit_equal(hp.frame(w).download(m), op.frame(w).map(m), f"final frame {w}.{name}")
    assert len(origins) >= 4, f"only {len(origins) - 1} origin shifts"
    assert max_map >= 1_000_000, f"map peaked at {max_map} surfels"
    assert cached_seen > 100_000, "no tiles were parked in the submap cache"


def synthetic_map(S, seed=7):
    """planar patches within +-88 m, as tools/stress_map.py (BASELINE config 5)"""
    rng = np.random.default_rng(seed)
    surf = np.zeros(S, dtype=SURFEL_DTYPE)
    xy = rng.uniform(-88, 88, (S, 2)).astype(np.float32)
    ground = rng.random(S) < 0.7
    surf["x"], surf["y"] = xy[:, 0], xy[:, 1]
    surf["z"] = np.where(ground, -1.73, rng.uniform(-1.7, 4.0, S)).astype(np.float32)
    ang = np.arctan2(-xy[:, 1], -xy[:, 0])
    surf["nx"] = np.where(ground, 0.0, np.cos(ang)).astype(np.float32)
    surf["ny"] = np.where(ground, 0.0, np.sin(ang)).astype(np.float32)
    surf["nz"] = np.where(ground, 1.0, 0.0).astype(np.float32)
    d = np.maximum(np.hypot(xy[:, 0], xy[:, 1]), 2.0)
    surf["radius"] = np.clip(1.41 * d * 0.0019, 0.03, 1.0).astype(np.float32)
    surf["confidence"] = rng.uniform(-1.0, 5.0, S).astype(np.float32)
    surf["timestamp"] = 0
    surf["count"] = 0.0
    surf["weight"] = 1.0
    surf["r"] = surf["g"] = surf["b"] = np.float32(40 / 255.0)
    surf["w"] = 0.9
    return surf


# BASELINE configs[4] names 50 M surfels: the literal size (9 s on the bench host with the 16-thread oracle), and 20 M
SURFEL_COUNTS = [20_000_000, 50_000_000]


@pytest.mark.parametrize("S", SURFEL_COUNTS, ids=lambda s: f"{s // 1_000_000}M")
def test_index_above_2_24(hip, oracle_lib, S):
    W, H = 4096, 128
    assert S > (1 << 24)
    p = params_with_size(W, H, max_surfels=S + 4 * W * H, cache_surfels=1 << 20)
    surf = synthetic_map(S)
    ctx = hip.Context(p)
    hmap = hip.SurfelMap(ctx)
    ora = oracle_lib.Oracle(p, threads=THREADS)
    hmap.upload(surf, 1)
    ora.map_upload(surf, 1)
    del surf
    pts, lab, prob, _ = get_scan(0, W, True, H)
    hf = hip.Frame(ctx, W, H)
    hip.Preprocessing(ctx).process(pts, hf, lab, prob, 20)
    of = ora.preprocess(pts, lab, prob, 20, ora.frame())
    pose = np.eye(4)
    # render of the uploaded map (K4 + K5), then one update (K7-K11) and the render after it
    for step in ("render", "update", "render"):
        if step == "update":
            hmap.update(pose, hf)
            ora.map_update(pose, of)
            him, oim = hmap.index_map(), ora.map_index_map()
            np.testing.assert_array_equal(him, oim, err_msg="index map")
            assert int(him.max()) > (1 << 24) and np.count_nonzero(him > (1 << 24)) > 1000, "no index beyond 2^24 in view"
            # a float index map would round these ids: the uint32 map must hold odd ids above 2^24
            assert np.count_nonzero((him > (1 << 24)) & (him % 2 == 1)) > 100
            np.testing.assert_array_equal(hmap.integrated(), ora.map_integrated(), err_msg="integration mask")
            assert_bit_equal(hmap.radius_conf(), ora.map_radius_conf(), "radius_conf")
            su, sn, _, _ = hmap.counts()
            assert (su, sn) == ora.map_counts()
            assert hmap.size() == ora.map_size()
            hs, os_ = hmap.getAllSurfels(), ora.map_surfels()
            assert hs.tobytes() == os_.tobytes(), "surfel buffers differ"
            del hs, os_
        else:
            hout, oout = hip.Frame(ctx, W, H), ora.frame(model=True)
            hmap.render(pose, pose, hout, 0.0)
            ora.map_render(pose, pose, 0.0, oout)
            for m, name in enumerate(("vertex", "normal", "semantic")):
                assert_bit_equal(hout.download(m), oout.map(m), f"{step} out.{name}")
            assert float((hout.download(0)[..., 3] > 0).mean()) > 0.3


def test_two_pipelines_one_gpu(hip, oracle_lib):
    W, N = 900, 12
    p = params_with_size(W)
    seqs = [[get_scan(k, W, True) for k in range(N)], [get_scan(2 * k, W, True) for k in range(N)]]

    def solo(seq):
        pipe = hip.SurfelMapping(p)
        poses = []
        for pts, lab, prob, _ in seq:
            pipe.processScan(pts, lab, prob, fixed_iterations=10)
            poses.append(pipe.getCurrentPose().copy())
        return poses, pipe.map.getAllSurfels().tobytes(), pipe.lastStats().as_dict()

    want = [solo(s) for s in seqs]
    pipes = [hip.SurfelMapping(p) for _ in seqs]
    got = [[], []]
    errors = []
    barrier = threading.Barrier(2)

    def run(i):
        try:
            for k, (pts, lab, prob, _) in enumerate(seqs[i]):
                if k % 3 == 0:
                    barrier.wait(timeout=60)  # keep the two streams interleaved
                pipes[i].processScan(pts, lab, prob, fixed_iterations=10)
                got[i].append(pipes[i].getCurrentPose().copy())
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for i in range(2):
        poses, surfels, stats = want[i]
        assert len(got[i]) == N
        for k in range(N):
            assert np.array_equal(got[i][k], poses[k]), f"pipeline {i} scan {k}: pose differs from the solo run"
        assert pipes[i].map.getAllSurfels().tobytes() == surfels, f"pipeline {i}: map differs from the solo run"
        assert pipes[i].lastStats().as_dict() == stats
    # and the solo run itself is the oracle's
    op = oracle_lib.OraclePipeline(p)
    for pts, lab, prob, _ in seqs[1]:
        op.process_scan(pts, lab, prob, fixed_iterations=10)
    assert np.array_equal(want[1][0][-1], op.pose()) and want[1][1] == op.ctx.map_surfels().tobytes()


def test_host_vector_entry_keeps_up_with_resident_scans(hip):
    """SurfelMapping::processScan(const rv::Laserscan&) as the reference's caller uses it (SurfelMapping.cpp:175,
    323-331): pageable host vectors, one blocking call per scan, no look-ahead -- against the same scans resident in
    HBM, both behind a 300-scan pre-roll (the steady ~1 M-surfel map bench.py times), 100 scans each, twice.  The entry
    stages through pinned memory and the copy stream while the previous scan's surfel passes run.  This is a PARITY
    test: both entries must produce the same bits.  The rate ratio is printed, never asserted -- it is a property of
    the host (round 5: 0.88-0.89 under the driver's cgroup, 0.98-1.01 elsewhere); bench.py's `host_vector_entry`
    object is where it is measured and broken down."""
    import time
    from semantic_suma_amd import synth
    W, PRE, N = 2048, 300, 100
    p = params_with_size(W)
    scans_ = [synth.generate_scan(k, n_azimuth=W)[:3] for k in range(PRE + 2 * N)]
    res, host = hip.SurfelMapping(p), hip.SurfelMapping(p)
    dev = [tuple(res.ctx.device_array(a) for a in sc) + (sc[0].shape[0],) for sc in scans_]
    for k in range(PRE):
        res.processScanDevice(*dev[k], fixed_iterations=10)
        if k < PRE - 3:
            host.processScanDevice(*dev[k], fixed_iterations=10)
        else:
            host.processScan(*scans_[k], fixed_iterations=10)  # staging buffers and copy threads exist before the clock starts
    best = 0.0
    for rep in range(2):
        lo, hi = PRE + rep * N, PRE + (rep + 1) * N
        # both stretches through the native host loop (suma_pipeline_run_scans), as bench.py drives them: an interpreter
        # between two calls costs the host entry more than the resident one (three arrays to marshal per scan) and was a
        # third of the 0.88 the round-5 driver run printed here
        job_r = res.prepareScans([dev[k] for k in range(lo, hi)], True)
        job_h = host.prepareScans([scans_[k] for k in range(lo, hi)], False)
        res.ctx.synchronize()
        t = time.perf_counter()
        assert res.runScans(job_r, True, fixed_iterations=10) == N
        res.ctx.synchronize()
        t_res = time.perf_counter() - t
        host.ctx.synchronize()
        host.hostEntryTimes(reset=True)
        t = time.perf_counter()
        assert host.runScans(job_h, False, fixed_iterations=10) == N
        host.ctx.synchronize()
        t_host = time.perf_counter() - t
        print(f"scans {lo}..{hi - 1}: resident {N / t_res:.0f} scans/s, host vectors {N / t_host:.0f} scans/s ({t_res / t_host:.3f}); "
              f"caller's time per call (us): {host.hostEntryTimes()}")
        best = max(best, t_res / t_host)
    assert np.array_equal(res.getCurrentPose(), host.getCurrentPose()), "both entries must run the same scans to the same bits"
    print(f"host-vector entry at {best:.2f} of the resident rate (reported, not asserted)")


def test_full_sequence_against_the_recorded_oracle_trace():
    """BASELINE configs[1] VERBATIM on the build under test: all 4541 scans (64 x 2048, semantic ICP, 10 GN iterations,
    up to 9 M surfels, 50 submap origins) through the HIP pipeline against the oracle's recorded trace
    (tests/golden/long_trace_4541.npz, made by `tools/long_parity.py --record` -- 20 minutes of CPU, once per oracle
    source; the trace names the oracle sources it belongs to and the check refuses another): pose bits, statistics and
    counters after EVERY scan, SHA-256 of the whole surfel buffer every 50 scans.  About a minute on the GPU box; the
    JSON it writes (kernel_source_sha + result) is what profiles/r05_long_parity_4541_scans.json holds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    trace = os.path.join(root, "tests", "golden", "long_trace_4541.npz")
    if not os.path.exists(trace):
        pytest.skip("tests/golden/long_trace_4541.npz not recorded")
    out = os.path.join(root, "gpurun_out", "long_parity.json")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "long_parity.py"), "--check", trace, "--out", out],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["result"] == "equal" and res["scans"] == 4541 and res["surfel_buffers_compared"] >= 90
    assert res["max_map_surfels"] > 8_000_000 and res["submap_origins_visited"] >= 40
