"""Host-scan hand-over: scans/s of the same 64x2048 sequence with (a) scans resident in HBM (bench.py's mode),
(b) the blocking host-pointer entry (suma_pipeline_process_scan: pageable H2D in front of every scan),
(c) the device-side ingest (suma_pipeline_prefetch_scan / process_prefetched: pinned double buffer + copy stream +
ingest thread, upload of scan k+1 overlapping the kernels of scan k)."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from semantic_suma_amd import core, synth
from semantic_suma_amd.types import params_with_size
W, K, Wu = 2048, 240, 10  # long enough that the three-scan head start of the staging is noise
p = params_with_size(W)
scans = [synth.generate_scan(k, n_azimuth=W)[:3] for k in range(Wu + K)]
out = {}
def run(name, fn):
    pipe = core.SurfelMapping(p)
    fn(pipe, scans[:Wu]); pipe.ctx.synchronize()
    t = time.perf_counter(); fn(pipe, scans[Wu:]); pipe.ctx.synchronize(); dt = time.perf_counter() - t
    out[name] = round(K / dt, 1); return pipe.getCurrentPose()
def resident(pipe, ss):
    dev = [(pipe.ctx.device_array(a), pipe.ctx.device_array(b), pipe.ctx.device_array(c), a.shape[0]) for a, b, c in ss]
    pipe.ctx.synchronize(); resident.t0 = time.perf_counter()
    for d in dev: pipe.processScanDevice(*d, fixed_iterations=10)
def blocking(pipe, ss):
    for a, b, c in ss: pipe.processScan(a, b, c, fixed_iterations=10)
def ingest(pipe, ss):
    pipe.processSequence(ss, fixed_iterations=10)
# resident: time only the scan loop (uploads are outside)
pipe = core.SurfelMapping(p); resident(pipe, scans[:Wu]); pipe.ctx.synchronize()
dev = [(pipe.ctx.device_array(a), pipe.ctx.device_array(b), pipe.ctx.device_array(c), a.shape[0]) for a, b, c in scans[Wu:]]
pipe.ctx.synchronize(); t = time.perf_counter()
for d in dev: pipe.processScanDevice(*d, fixed_iterations=10)
pipe.ctx.synchronize(); out["resident_hbm"] = round(K / (time.perf_counter() - t), 1); P0 = pipe.getCurrentPose()
P1 = run("blocking_host_pointer", blocking)
P2 = run("async_ingest", ingest)
assert np.array_equal(P0, P1) and np.array_equal(P0, P2)
out["async_vs_resident"] = round(out["async_ingest"] / out["resident_hbm"], 3)
print(json.dumps(out))
