"""Feasibility probe (round 4): how much do a Gauss-Newton chain and a surfel render pass cost each other when they run
SIDE BY SIDE on one GPU?  The post-ICP render_active + resolve + statistics pass of scan t feed neither the pose nor the
map (only lastModelFrame and result_new_, SurfelMapping.cpp:406-423): they could run on a stream of their own under the
Gauss-Newton chain of scan t + 1, which leaves three quarters of the chip idle.  Two contexts, two host threads: one
loops LieGaussNewton::minimize (10 iterations), the other SurfelMap::render_active on a map of the steady size."""
import sys, threading, time
import numpy as np
sys.path.insert(0, '.')
from semantic_suma_amd import core, synth
from semantic_suma_amd.types import params_with_size

W, WARM, N = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 200, 150
p = params_with_size(W, max_iterations=10, stopping_threshold=0.0, delta=0.0)
pa, pb = core.SurfelMapping(p), core.SurfelMapping(p)
for k in range(WARM):
    pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W)
    pa.processScan(pts, lab, prob, fixed_iterations=10)
    pb.processScan(pts, lab, prob, fixed_iterations=10)
pa.ctx.synchronize(); pb.ctx.synchronize()
print(f"maps: {pa.map.size()} / {pb.map.size()} surfels")
pose = pa.getCurrentPose()
obj, gn = core.Frame2Model(pa.ctx), core.LieGaussNewton(pa.ctx)
obj.setData(pa.frame(0), pa.map.newMapFrame())


def gn_loop(n, out):
    t = time.perf_counter()
    for _ in range(n):
        gn.minimize(obj, np.eye(4))
    out["gn"] = (time.perf_counter() - t) / n * 1e6


def render_loop(n, out, stop=None):
    t = time.perf_counter()
    k = 0
    while (k < n) if stop is None else (not stop.is_set()):
        pb.map.render_active(pose, 0.0)
        k += 1
        if k % 8 == 0:
            pb.ctx.synchronize()
    pb.ctx.synchronize()
    out["render"] = (time.perf_counter() - t) / max(k, 1) * 1e6


solo = {}
gn_loop(20, {}); render_loop(20, {})
gn_loop(N, solo); render_loop(N, solo)
both, stop = {}, threading.Event()
th = threading.Thread(target=render_loop, args=(0, both, stop))
th.start()
time.sleep(0.01)
gn_loop(N, both)
stop.set(); th.join()
print(f"alone:    minimize (10 GN iterations + closing launch, host round trip) {solo['gn']:.1f} us   render_active + resolve {solo['render']:.1f} us")
print(f"together: minimize {both['gn']:.1f} us ({100 * (both['gn'] / solo['gn'] - 1):+.1f} %)   render_active + resolve {both['render']:.1f} us ({100 * (both['render'] / solo['render'] - 1):+.1f} %)")
