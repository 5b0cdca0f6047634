#!/usr/bin/env python
"""Why was ONE host-vector call 3 - 6 ms long in two bench records (BENCH_r04: 0.67 of resident on 20 scans; round 5's
default run: first call of the stretch 6.3 ms)?  Both times the stretch came right behind a stretch of RESIDENT scans,
i.e. the copy stream and the pinned staging had been idle for 7 - 19 ms.  This probe measures the first host-vector call
after T ms during which the pipeline ran resident scans only (copy path idle), and after T ms of complete idleness:
    python tools/host_entry_idle_probe.py   ->  one JSON line (gpurun_out/host_entry_idle_probe.json)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_suma_amd import core, synth  # noqa: E402
from semantic_suma_amd.types import params_with_size  # noqa: E402

W, H = 2048, 64
p = params_with_size(W, H)
pipe = core.SurfelMapping(p)
N = 420
scans = [synth.generate_scan(k % 300, n_azimuth=W, height=H)[:3] for k in range(60)]
dev = [tuple(pipe.ctx.device_array(a) for a in sc) + (sc[0].shape[0],) for sc in scans]
k = 0


def host(n):
    global k
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        pipe.processScan(*scans[k % 60], fixed_iterations=10)
        ts.append((time.perf_counter() - t) * 1e6)
        k += 1
    return ts


def resident_for(ms):
    global k
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        pipe.processScanDevice(*dev[k % 60], fixed_iterations=10)
        k += 1
    pipe.ctx.synchronize()


for _ in range(40):  # a map of some size, the entry warmed
    host(1)
out = {"after_resident_ms": {}, "after_idle_ms": {}}
for T in (0, 2, 5, 10, 20, 50, 100, 300):
    r = []
    for rep in range(3):
        host(5)
        resident_for(T)
        r.append(round(host(1)[0], 1))
    out["after_resident_ms"][T] = r
    r = []
    for rep in range(3):
        host(5)
        pipe.ctx.synchronize()
        time.sleep(T / 1e3)
        r.append(round(host(1)[0], 1))
    out["after_idle_ms"][T] = r
out["steady_call_us_median"] = round(float(np.median(host(30))), 1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "host_entry_idle_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
