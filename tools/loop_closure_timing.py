"""where a loop-closure verification (SurfelMapping.cpp:679-757) spends its time on the device, batched
(suma_loop_closure_verify) against the reference's one-by-one sequencing (suma_loop_closure_verify_serial) and against a
single guess: wall clock per verification and the HIP-event time of each kernel group, on a map of `scans` scans."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import get_scan
from semantic_suma_amd import core
from semantic_suma_amd.types import params_with_size
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 150
max_it = int(sys.argv[3]) if len(sys.argv) > 3 else 33
p = params_with_size(W, max_iterations=max_it)
pipe = core.SurfelMapping(p)
ctx = pipe.ctx
for k in range(scans):
    pts, lab, prob, _ = get_scan(k, W)
    pipe.processScan(pts, lab, prob, fixed_iterations=10)
cur = pipe.frame(0)
prior = pipe.getCurrentPose().astype(np.float64)
Rz = np.eye(4); Rz[:2, :2] = -Rz[:2, :2]
half = np.eye(4); half[0, 3] = 0.5
inits = [np.eye(4), Rz, half]
ct = float(p.confidence_threshold)
out = {"width": W, "map_surfels": pipe.map.size(), "max_iterations": max_it, "n_init": 3, "unit": "us"}
for gname, gates in (("reference_gates", (0.2, 0.85)), ("no_guess_passes", (2.0, 0.85)), ("every_guess_passes", (-1.0, 2.0))):
    row = {}
    for name, guesses, serial in (("batched", inits, False), ("serial", inits, True), ("one_guess", inits[:1], False)):
        core.loop_closure_verify(ctx, cur, prior, guesses, prior, ct, *gates, serial=serial)
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            r = core.loop_closure_verify(ctx, cur, prior, guesses, prior, ct, *gates, serial=serial)
        ctx.synchronize()
        wall = 1e6 * (time.perf_counter() - t) / 20
        ctx.profile(1); ctx.profile_reset()
        for _ in range(5):
            core.loop_closure_verify(ctx, cur, prior, guesses, prior, ct, *gates, serial=serial)
        groups = {q["name"]: [q["launches"] // 5, round(1000 * q["total_ms"] / 5, 1)] for q in ctx.profile_get() if q["launches"]}
        ctx.profile(0)
        row[name] = {"wall": round(wall, 1), "passed": [bool(x["passed"]) for x in r],
                     "iterations": [x["after_minimize"]["iterations"] for x in r], "groups_launches_us": groups}
    row["batched_over_one_guess"] = round(row["batched"]["wall"] / row["one_guess"]["wall"], 3)
    row["serial_over_batched"] = round(row["serial"]["wall"] / row["batched"]["wall"], 3)
    out[gname] = row
print(json.dumps(out, indent=1))
