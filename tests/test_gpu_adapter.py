"""GPU tests of the class-by-class boundary (run with -m gpu on an MI355X): what the reference's untouched
SurfelMapping.cpp calls one method at a time -- Preprocessing::process, SurfelMap::render / render_active / update,
Frame2Model::jacobianProducts, LieGaussNewton::minimize -- must give the bits of the scan pipeline (which the parity
suite pins to the oracle) although round 4 fuses work across those calls: the scan upload + K1-K3 on the side stream
behind a deferred wait, render() de-duplicated for caller-owned frames (frame versions), the index-map splat (K7)
speculated into render_active, K8 riding on jacobianProducts, results polled from pinned records."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_bit_equal, get_scan
from semantic_suma_amd.types import params_with_size

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 900


@pytest.fixture(scope="module")
def hip():
    from semantic_suma_amd import core
    core.lib()
    return core


def conf_threshold(p, t):
    """SurfelMapping::getConfidenceThreshold, SurfelMapping.cpp:333-340"""
    ct = np.float32(p.confidence_threshold)
    if t < 10:
        log_unstable = np.float32(np.log(np.float64(np.float32(0.1) / (np.float32(1.0) - np.float32(0.1)))))
        alpha = np.float32(t) / np.float32(10)
        ct = np.float32((1.0 - np.float64(alpha)) * np.float64(log_unstable) + np.float64(alpha * np.float32(p.confidence_threshold)))
    return float(ct)


def mul4(A, B):
    C = np.zeros((4, 4))
    for r in range(4):
        for c in range(4):
            C[r, c] = ((A[r, 0] * B[0, c] + A[r, 1] * B[1, c]) + A[r, 2] * B[2, c]) + A[r, 3] * B[3, c]
    return C


class ClassPath:
    """SurfelMapping::processScan without loop closures on the mirror classes, call by call
    (SurfelMapping.cpp:175-210, 323-358, 372-476, 797-804; the same sequence as tools/adapter_bench.cpp)"""

    def __init__(self, hip, p, iterations):
        self.hip, self.p = hip, p
        self.ctx = hip.Context(p)
        self.pre, self.map, self.gn = hip.Preprocessing(self.ctx), hip.SurfelMap(self.ctx), hip.LieGaussNewton(self.ctx)
        self.objective = hip.Frame2Model(self.ctx)
        self.current, self.last = hip.Frame(self.ctx, p.data_width, p.data_height), hip.Frame(self.ctx, p.data_width, p.data_height)
        self.current_model = hip.Frame(self.ctx, p.model_width, p.model_height)
        self.last_model = hip.Frame(self.ctx, p.model_width, p.model_height)
        self.pose, self.increment, self.t = np.eye(4), np.eye(4), 0
        self.stats = None

    def process(self, pts, lab, prob):
        p, k = self.p, self.t
        self.current, self.last = self.last, self.current
        self.current_model, self.last_model = self.last_model, self.current_model
        self.pre.process(pts, self.current, lab, prob, k)
        self.map.render(self.pose, self.pose, self.last_model, conf_threshold(p, k))
        if k > 0:
            self.objective.setData(self.current, self.map.newMapFrame())
            self.gn.minimize(self.objective, self.increment)
            inc = self.gn.pose().copy()
            posed = mul4(self.pose, inc)
            self.map.render_active(posed, conf_threshold(p, k))
            self.last_model.copy(self.map.newMapFrame())
            self.objective.setData(self.current, self.map.newMapFrame())
            self.objective.initialize(np.eye(4))
            self.objective.jacobianProducts()
            self.stats = (self.objective.valid(), self.objective.outlier(), self.objective.invalid())
            self.pose = posed
            self.increment = inc
        self.map.update(self.pose, self.current)
        self.map.render(self.pose, self.pose, self.current_model, conf_threshold(p, k))
        self.t += 1


def test_class_by_class_sequence_equals_the_pipeline(hip):
    """poses, statistics, index map, integration mask, K8 products and the whole surfel buffer after every scan"""
    p = params_with_size(W, max_iterations=10, stopping_threshold=0.0, delta=0.0)
    cp = ClassPath(hip, p, 10)
    pipe = hip.SurfelMapping(p)
    for k in range(7):
        pts, lab, prob, _ = get_scan(k, W, True)
        cp.process(pts, lab, prob)
        pipe.processScan(pts, lab, prob, fixed_iterations=10)
        assert np.array_equal(cp.pose, pipe.getCurrentPose()), f"scan {k}: pose"
        np.testing.assert_array_equal(cp.map.index_map(), pipe.map.index_map(), err_msg=f"scan {k}: index map")
        np.testing.assert_array_equal(cp.map.integrated(), pipe.map.integrated(), err_msg=f"scan {k}: integration mask")
        assert_bit_equal(cp.map.radius_conf(), pipe.map.radius_conf(), f"scan {k}: radius_conf")
        assert cp.map.counts() == pipe.map.counts(), f"scan {k}: counts"
        assert cp.map.getAllSurfels().tobytes() == pipe.map.getAllSurfels().tobytes(), f"scan {k}: surfels"
        if k > 0:
            st = pipe.lastStats()
            assert cp.stats == (st.valid, st.outlier, st.invalid), f"scan {k}: statistics pass"
        for m in range(3):
            assert_bit_equal(cp.current_model.download(m), pipe.frame(2).download(m), f"scan {k}: model frame map {m}")


def launches(ctx, name):
    return sum(k["launches"] for k in ctx.profile_get() if k["name"] == name)


def test_render_is_deduplicated_for_caller_owned_frames(hip):
    """SurfelMapping renders the same map from the same pose at the end of scan t (:803) and at the start of scan t + 1
    (:351) into the frame the shared_ptr swap hands over: the second call launches nothing -- unless something wrote one
    of the three targets in between, which every writing call records in the frame's version"""
    p = params_with_size(W)
    ctx = hip.Context(p)
    pre, smap = hip.Preprocessing(ctx), hip.SurfelMap(ctx)
    f = hip.Frame(ctx, W, 64)
    pts, lab, prob, _ = get_scan(0, W, True)
    pre.process(pts, f, lab, prob, 0)
    smap.update(np.eye(4), f)
    out, other = hip.Frame(ctx, W, 64), hip.Frame(ctx, W, 64)
    ctx.profile(1)
    ctx.profile_reset()
    smap.render(np.eye(4), np.eye(4), out, -2.0)
    assert launches(ctx, "k4_render_surfels") == 1
    want = [out.download(m) for m in range(3)]
    assert want[0][..., 3].sum() > 1000
    smap.render(np.eye(4), np.eye(4), out, -2.0)
    assert launches(ctx, "k4_render_surfels") == 1, "identical call: no launch"
    # every kind of write to one of the targets brings the launch back, and the result is the same bits
    writers = [lambda: out.upload(0, np.zeros_like(want[0])), lambda: out.copy(other), lambda: out.swap(other),
               lambda: out.swap(other), lambda: pre.process(pts, out, lab, prob, 0),
               lambda: smap.render_active(np.eye(4), -2.0), lambda: smap.render(np.eye(4), np.eye(4), other, -2.0),
               lambda: smap.newMapFrame().upload(1, np.zeros_like(want[0])), lambda: out.touch()]
    n = 1
    for i, wr in enumerate(writers):
        wr()
        smap.render(np.eye(4), np.eye(4), out, -2.0)
        n += 1 + (1 if i in (5, 6) else 0)  # the writers that render launch k_render themselves
        assert launches(ctx, "k4_render_surfels") + launches(ctx, "k4k7_render_indexmap") == n, f"writer {i}"
        for m in range(3):
            assert_bit_equal(out.download(m), want[m], f"after writer {i}: map {m}")
    # other pose / threshold: a real render
    T = np.eye(4)
    T[0, 3] = 0.25
    smap.render(T, T, out, -2.0)
    smap.render(np.eye(4), np.eye(4), out, 0.5)
    assert launches(ctx, "k4_render_surfels") + launches(ctx, "k4k7_render_indexmap") == n + 2
    ctx.profile(0)


def test_index_map_speculation_in_render_active(hip, oracle_lib):
    """render_active splats the index map for the update that follows at the same pose; an update at ANOTHER pose, a
    second render_active, or new parameters in between must not see that splat (checked against the oracle)"""
    p = params_with_size(W)
    ctx, ora = hip.Context(p), oracle_lib.Oracle(p)
    pre, smap = hip.Preprocessing(ctx), hip.SurfelMap(ctx)
    T0 = get_scan(0, W, True)[3]
    ctx.profile(1)
    ctx.profile_reset()
    k7_alone = 0
    for i in range(9):
        pts, lab, prob, T = get_scan(i, W, True)
        pose = np.linalg.inv(T0) @ T
        hf = hip.Frame(ctx, W, 64)
        pre.process(pts, hf, lab, prob, i)
        of = ora.preprocess(pts, lab, prob, i, ora.frame())
        off = pose.copy()
        off[0, 3] += 0.3
        if i in (0, 1, 5, 7, 8):   # the reference's sequence: render_active(pose) -> update(pose): splat consumed ...
            smap.render_active(pose, -2.0)
            if i == 7:             # ... unless the speculation is off (after scans 5 / 6): this match turns it on again
                k7_alone += 1
        elif i == 2:               # update at another pose than the one rendered: the splat is dropped
            smap.render_active(off, -2.0)
            k7_alone += 1
        elif i == 3:               # speculation is off now; this update WOULD have matched: it comes back on
            smap.render_active(pose, -2.0)
            k7_alone += 1
        elif i == 4:               # two active renders in a row, then parameters re-sent: dropped
            smap.render_active(pose, -2.0)
            smap.render_active(pose, -2.0)
            k7_alone += 1
        elif i == 6:               # no active render at all
            k7_alone += 1
        if i == 5:
            ctx.set_params(p)      # params_version changes between the splat and the update
            k7_alone += 1
        smap.update(pose, hf)
        ora.map_update(pose, of)
        np.testing.assert_array_equal(smap.index_map(), ora.map_index_map(), err_msg=f"scan {i}: index map")
        assert smap.getAllSurfels().tobytes() == ora.map_surfels().tobytes(), f"scan {i}: surfels"
    assert launches(ctx, "k7_indexmap") == k7_alone
    ctx.profile(0)


def test_jacobian_products_carries_k8_once(hip, oracle_lib):
    """Frame2Model::jacobianProducts through the C-ABI is ONE self-closing launch that reports through a pinned record;
    on a data-sized frame it also forms the K8 products the following update needs -- once per frame contents"""
    p = params_with_size(W)
    ctx, ora = hip.Context(p), oracle_lib.Oracle(p)
    pre, smap, obj = hip.Preprocessing(ctx), hip.SurfelMap(ctx), hip.Frame2Model(ctx)
    frames, oframes = [], []
    for i in range(2):
        pts, lab, prob, _ = get_scan(i, W, True)
        f = hip.Frame(ctx, W, 64)
        pre.process(pts, f, lab, prob, i)
        frames.append(f)
        oframes.append(ora.preprocess(pts, lab, prob, i, ora.frame()))
    smap.update(np.eye(4), frames[0])
    ora.map_update(np.eye(4), oframes[0])
    model, omodel = hip.Frame(ctx, W, 64), ora.frame(model=True)
    smap.render(np.eye(4), np.eye(4), model, -2.0)
    ora.map_render(np.eye(4), np.eye(4), -2.0, omodel)
    obj.setData(frames[1], smap.newMapFrame())
    ctx.profile(1)
    ctx.profile_reset()
    for rep in range(3):
        obj.initialize(np.eye(4))
        F, JtJ, Jtr = obj.jacobianProducts()
        Fo, acco, JtJo, Jtro, sto = ora.jacobian_products(oframes[1], ora.map_frame(1), np.eye(4), 0)
        assert F == Fo and np.array_equal(JtJ, JtJo.T) and np.array_equal(Jtr, Jtro), f"call {rep}"
        assert np.array_equal(obj.acc, acco)
        assert (obj.valid(), obj.outlier(), obj.invalid()) == (sto.valid, sto.outlier, sto.invalid)
        assert np.array_equal(hip.LieGaussNewton(ctx).information(), JtJ)
    assert launches(ctx, "k6k8_stats_radius") == 1 and launches(ctx, "k6_icp_step") == 2  # K8 rode along once
    assert launches(ctx, "k6_icp_finish") == 0
    smap.update(np.eye(4), frames[1])
    ora.map_update(np.eye(4), oframes[1])
    assert launches(ctx, "k8_radius") == 0, "the update took the fused products"
    assert_bit_equal(smap.radius_conf(), ora.map_radius_conf(), "radius_conf")
    assert smap.getAllSurfels().tobytes() == ora.map_surfels().tobytes()
    # a frame whose contents changed after the fused pass: the products are stale and the update makes its own
    pts, lab, prob, _ = get_scan(2, W, True)
    obj.setData(frames[0], smap.newMapFrame())
    obj.initialize(np.eye(4))
    obj.jacobianProducts()
    pre.process(pts, frames[0], lab, prob, 2)
    of2 = ora.preprocess(pts, lab, prob, 2, ora.frame())
    smap.update(np.eye(4), frames[0])
    ora.map_update(np.eye(4), of2)
    assert launches(ctx, "k8_radius") == 1
    assert smap.getAllSurfels().tobytes() == ora.map_surfels().tobytes()
    ctx.profile(0)


def test_preprocess_waits_only_when_the_frame_is_still_in_use(hip, oracle_lib):
    """Preprocessing::process from host vectors runs upload + K1-K3 on the side stream; every kind of reader that
    follows must see the finished frame, and a frame that ctx-stream work enqueued just before still reads must not be
    overwritten under it"""
    p = params_with_size(W)
    ctx, ora = hip.Context(p), oracle_lib.Oracle(p)
    pre = hip.Preprocessing(ctx)
    f, g = hip.Frame(ctx, W, 64), hip.Frame(ctx, W, 64)
    for i in range(6):
        pts, lab, prob, _ = get_scan(i, W, True)
        of = ora.preprocess(pts, lab, prob, i, ora.frame())
        pre.process(pts, f, lab, prob, i)
        if i % 3 == 0:      # reader: download
            pass
        elif i % 3 == 1:    # reader: Frame::copy, then overwrite the source at once (WAR on f)
            g.copy(f)
            pts2, lab2, prob2, _ = get_scan(i + 10, W, True)
            pre.process(pts2, f, lab2, prob2, i)
            for m in range(3):
                assert_bit_equal(g.download(m), of.map(m), f"scan {i}: copy taken before the overwrite, map {m}")
            continue
        else:               # reader: swap, the data must travel with the maps
            f.swap(g)
            f, g = g, f
        for m in range(3):
            assert_bit_equal(f.download(m), of.map(m), f"scan {i}: map {m}")


def test_six_pipelines_on_two_hardware_queues():
    """suma_run_sequences with six concurrent pipelines (x three streams each) while the runtime has TWO hardware queues:
    the in-memory gate would let pipelines block each other's producers (round-3 review); with more than one pipeline in
    the process the hand-off is a runtime event dependency, and the run must finish with every sequence's poses equal to
    a one-at-a-time run"""
    code = f"""
import sys
sys.path.insert(0, {ROOT!r})
import numpy as np
from semantic_suma_amd import synth
from semantic_suma_amd.distributed import run_sequences_hip
from semantic_suma_amd.types import params_with_size
p = params_with_size({W}, max_surfels=400000)
seqs = {{s: [synth.generate_scan(100 * s + k, n_azimuth={W})[:3] for k in range(7)] for s in range(6)}}
many = run_sequences_hip(p, seqs, fixed_iterations=6, max_concurrent=6)
one = run_sequences_hip(p, seqs, fixed_iterations=6, max_concurrent=1)
for s in seqs:
    assert many[s][0] == 7 and np.array_equal(many[s][1], one[s][1]), s
print("SIX_OK")
"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "SIX_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_hypothesis_runner_reports_capacity(hip):
    """the hypothesis path never runs update_pose, whose result record carries DevState: the overflow bits must reach
    the host all the same (round-3 advisor)"""
    from semantic_suma_amd.distributed import run_hypotheses_hip
    p = params_with_size(W, max_surfels=60000)
    scans = [get_scan(k, W, True)[:3] for k in range(6)]
    with pytest.raises(hip.SumaError, match="capacity"):
        run_hypotheses_hip(p, scans, 4, fixed_iterations=5)
