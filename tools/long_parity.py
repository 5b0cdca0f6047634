"""One-off: N scans of the bench workload (64x2048, semantic ICP, 10 iterations) through the HIP pipeline and
the CPU oracle; pose bits, statistics and the whole surfel buffer compared after every scan."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from semantic_suma_amd import core, synth
from semantic_suma_amd.types import params_with_size
from oracle import pyoracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W = 2048
p = params_with_size(W)
hp, op = core.SurfelMapping(p), pyoracle.OraclePipeline(p)
t0 = time.time()
for k in range(N):
    pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W)
    hp.processScan(pts, lab, prob, fixed_iterations=10)
    op.process_scan(pts, lab, prob, fixed_iterations=10)
    assert np.array_equal(hp.getCurrentPose(), op.pose()), f"scan {k}: pose"
    assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k}: stats"
    if k % 5 == 4 or k == N - 1:
        assert hp.map.getAllSurfels().tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k}: surfels"
        print(f"scan {k}: map {hp.map.size()} surfels, origin {hp.map.counts()[3]}, cached {hp.map.counts()[2]}, x = {hp.getCurrentPose()[0,3]:.3f}  [{time.time()-t0:.0f} s]", flush=True)
print("long parity ok")
