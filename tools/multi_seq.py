"""Experiment: M independent sequences on ONE GPU (one pipeline + stream per sequence, one host thread each).
The single-sequence pipeline is latency bound; concurrent sequences fill the idle CUs."""
import sys, time, threading
import numpy as np
sys.path.insert(0, '.')
from semantic_suma_amd import core, synth
from semantic_suma_amd.types import params_with_size
M = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W = 2048
p = params_with_size(W)
scans = [synth.generate_scan(k, n_azimuth=W) for k in range(K + 5)]
pipes = [core.SurfelMapping(p) for _ in range(M)]
dev = [[(pl.ctx.device_array(s[0]), pl.ctx.device_array(s[1]), pl.ctx.device_array(s[2]), s[0].shape[0]) for s in scans] for pl in pipes]
def run(i, lo, hi):
    for k in range(lo, hi):
        pipes[i].processScanDevice(*dev[i][k], fixed_iterations=10)
    pipes[i].ctx.synchronize()
for i in range(M): run(i, 0, 5)
t = time.perf_counter()
th = [threading.Thread(target=run, args=(i, 5, 5 + K)) for i in range(M)]
[x.start() for x in th]; [x.join() for x in th]
dt = time.perf_counter() - t
print(f"M={M}: {M * K / dt:.0f} scans/s aggregate ({K / dt:.0f} per sequence), pose x = {pipes[0].getCurrentPose()[0,3]:.3f} / {pipes[-1].getCurrentPose()[0,3]:.3f}")
