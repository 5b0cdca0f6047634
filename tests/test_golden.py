"""Golden-vector tests: tests/golden/pipeline_360x32.npz (made by tests/golden/make_golden.py from the
CPU oracle; regression pins, not reference pins -- the reference has no fixtures).
CPU: the oracle reproduces them.  GPU (-m gpu): the HIP path reproduces them through the C-ABI."""
import hashlib
import os

import numpy as np
import pytest

from semantic_suma_amd.types import params_with_size

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_360x32.npz"))
W, H, N = int(G["W"]), int(G["H"]), int(G["n_scans"])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_scan(k, vertex, normal, semantic, pose, surfels, counts, st, model_vertex):
    assert sha(vertex) == str(G[f"sha_vertex{k}"]) and sha(normal) == str(G[f"sha_normal{k}"])
    assert sha(semantic) == str(G[f"sha_semantic{k}"])
    assert np.array_equal(pose, G[f"pose{k}"]), f"scan {k} pose"
    assert len(surfels) == int(G[f"map_size{k}"]) and sha(surfels) == str(G[f"sha_map{k}"])
    assert tuple(counts) == tuple(G[f"counts{k}"])
    assert [st.valid, st.outlier, st.inlier, st.invalid, st.iterations] == list(G[f"stats{k}"])
    assert [st.error, st.inlier_residual] == list(G[f"stats_f{k}"])
    assert sha(model_vertex) == str(G[f"sha_model{k}"])


def test_oracle_reproduces_golden(oracle_lib):
    p = params_with_size(W, H)
    pipe = oracle_lib.OraclePipeline(p)
    for k in range(N):
        pipe.process_scan(G[f"pts{k}"], G[f"lab{k}"], G[f"prob{k}"], fixed_iterations=10)
        f = pipe.frame(0)
        check_scan(k, f.vertex, f.normal, f.semantic, pipe.pose(), pipe.ctx.map_surfels(), pipe.ctx.map_counts(),
                   pipe.last_stats(), pipe.frame(2).vertex)
    ora = oracle_lib.Oracle(p)
    f0 = ora.preprocess(G["pts0"], G["lab0"], G["prob0"], 20, ora.frame())
    f1 = ora.preprocess(G["pts1"], G["lab1"], G["prob1"], 21, ora.frame())
    assert np.array_equal(ora.jacobian_products(f1, f0, G["k6_T"], 0)[1], G["k6_acc"])


@pytest.mark.gpu
def test_hip_reproduces_golden():
    from semantic_suma_amd import core
    p = params_with_size(W, H)
    pipe = core.SurfelMapping(p)
    for k in range(N):
        pipe.processScan(G[f"pts{k}"], G[f"lab{k}"], G[f"prob{k}"], fixed_iterations=10)
        f = pipe.frame(0)
        su, sn, _, _ = pipe.map.counts()
        check_scan(k, f.download(0), f.download(1), f.download(2), pipe.getCurrentPose(), pipe.map.getAllSurfels(),
                   (su, sn), pipe.lastStats(), pipe.frame(2).download(0))
    ctx = core.Context(p)
    pre = core.Preprocessing(ctx)
    f0, f1 = core.Frame(ctx, W, H), core.Frame(ctx, W, H)
    pre.process(G["pts0"], f0, G["lab0"], G["prob0"], 20)
    pre.process(G["pts1"], f1, G["lab1"], G["prob1"], 21)
    obj = core.Frame2Model(ctx)
    obj.setData(f1, f0)
    obj.initialize(G["k6_T"])
    obj.jacobianProducts()
    assert np.array_equal(obj.acc, G["k6_acc"])


def test_recorded_full_sequence_trace_belongs_to_this_oracle(oracle_lib):
    """tests/golden/long_trace_4541.npz is what the GPU suite checks the literal BASELINE configs[1] run against
    (tests/test_gpu_long.py, tools/long_parity.py --check).  It must have been recorded on THESE oracle sources -- the hash
    in the file is compared, and the first 50 scans are replayed here (pose bits, statistics, counters, and the surfel
    buffer's SHA-256 where the trace holds one): a stale trace fails on the build machine, not on the GPU box."""
    import hashlib
    import importlib.util
    import os
    from semantic_suma_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "long_trace_4541.npz")
    if not os.path.exists(path):
        pytest.skip("trace not recorded")
    spec = importlib.util.spec_from_file_location("long_parity", os.path.join(root, "tools", "long_parity.py"))
    lp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lp)
    z = np.load(path)
    assert str(z["oracle_source_sha"]) == lp.oracle_source_sha(), "re-record: python tools/long_parity.py --record tests/golden/long_trace_4541.npz"
    assert int(z["scans"]) == 4541 and z["poses"].shape == (4541, 4, 4) and len(z["sha_idx"]) >= 90
    W, H = int(z["width"]), int(z["height"])
    p = params_with_size(W, H, max_surfels=int(z["max_surfels"]))
    op = oracle_lib.OraclePipeline(p, threads=max(1, min(8, os.cpu_count() or 1)))
    keys = [str(k) for k in z["stat_keys"]]
    sha_at = {int(k): i for i, k in enumerate(z["sha_idx"])}
    every = int(z["every"])
    for k in range(every):  # up to and including the first surfel-buffer checkpoint
        pts, lab, prob = synth.generate_scan(k, n_azimuth=W, height=H)[:3]
        op.process_scan(pts, lab, prob, fixed_iterations=10)
        assert np.array_equal(op.pose(), z["poses"][k]), f"scan {k}: pose bits"
        st = op.last_stats().as_dict()
        assert [float(st[q]) for q in keys] == list(z["stats"][k]), f"scan {k}: statistics"
        assert (*op.ctx.map_counts(), op.ctx.map_cached_surfels(), *op.ctx.map_submap_origin()) == tuple(int(v) for v in z["counts"][k])
        if k in sha_at:
            assert hashlib.sha256(op.ctx.map_surfels().tobytes()).digest() == z["sha"][sha_at[k]].tobytes()
