#!/usr/bin/env python
"""The literal CPU baseline of north_star -- "the reference's own path timed on the host CPU (core count stated)":
SurfelMapping::processScan with every GL pass executed by the reference's OWN GLSL (read from /root/reference/src/shader
where it lies) in Mesa llvmpipe, host code between the passes in numpy (oracle/glpipeline.py), on the bench workload:
64 x 2048, 10 Gauss-Newton iterations, the steady-state map.  The oracle (16 threads, seconds) replays the pre-roll and
hands its map, pose table and poses to the GL pipeline; then --scans scans are timed.  CONTEXT ONLY (a software rasteriser
is not how anybody runs the reference), never credit: profiles/r05_reference_glsl_on_llvmpipe.json.
    python tools/reference_gl_baseline.py [--preroll 300] [--scans 10] [--out profiles/...json]
bench.py calls measure() for its optional `cpu_baseline_reference_gl` key when Mesa and the shaders are present."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, ITER = 2048, 64, 10


def measure(preroll=300, scans=10, threads=None, get_scan=None, log=None):
    from oracle import glref, pyref
    if not (glref.available() and pyref.available()):
        return None
    from oracle import glpipeline, pyoracle
    from semantic_suma_amd import synth
    from semantic_suma_amd.types import params_with_size
    get_scan = get_scan or (lambda k: synth.generate_scan(k, n_azimuth=W, height=H)[:3])
    threads = threads or max(1, min(16, os.cpu_count() or 1))
    p = params_with_size(W, H)
    op = pyoracle.OraclePipeline(p, threads=threads)
    t0 = time.perf_counter()
    for k in range(preroll):
        op.process_scan(*get_scan(k), fixed_iterations=ITER)
    t_pre = time.perf_counter() - t0
    g = glpipeline.GLPipeline(p)
    if preroll:
        ctx, ts = op.ctx, op.ctx.map_timestamp()
        g.surfels = np.ascontiguousarray(ctx.map_surfels()).view(np.float32).reshape(-1, 16).copy()
        g.poses[: ts + 1] = ctx.map_poses(ts + 1).reshape(-1, 16)
        g.timestamp, g.origin = ts, list(ctx.map_submap_origin())
        g.current_pose, g.last_increment = op.pose().copy(), op.last_increment().copy()
        f = op.frame(0)
        g.frame = (f.vertex.copy(), f.normal.copy(), f.semantic.copy())
    n0 = g.surfels.shape[0]
    per = []
    for k in range(preroll, preroll + scans):
        sc = get_scan(k)
        t = time.perf_counter()
        g.process_scan(*sc, ITER)
        per.append(time.perf_counter() - t)
        op.process_scan(*sc, fixed_iterations=ITER)
        if log:
            log(f"scan {k}: {per[-1]:.2f} s in GL, map {g.surfels.shape[0]} (oracle {op.ctx.map_size()})")
    D = np.linalg.inv(op.pose()) @ g.current_pose
    info = glref.limits()
    return {"what": "SurfelMapping::processScan with the reference's own GLSL (src/shader) executed by a real OpenGL on the host "
                    "CPU; host code between the passes in numpy (oracle/glpipeline.py)",
            "value": scans / sum(per), "unit": "scans/s", "kind": "reference", "cores": os.cpu_count(),
            "gl_renderer": info["renderer"], "gl_version": info["version"],
            "sample": f"scans {preroll}..{preroll + scans - 1} of the bench sequence ({H}x{W}, {ITER} GN iterations) on the map the "
                      f"oracle built over scans 0..{preroll - 1} ({n0} surfels; {t_pre:.1f} s, {threads} threads)",
            "seconds_per_scan": [round(x, 3) for x in per], "map_surfels_start": int(n0), "map_surfels_end": int(g.surfels.shape[0]),
            "oracle_map_surfels_end": int(op.ctx.map_size()),
            "end_pose_vs_oracle_m": float(np.linalg.norm(D[:3, 3]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preroll", type=int, default=300)
    ap.add_argument("--scans", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = measure(a.preroll, a.scans, log=lambda m: print(m, file=sys.stderr, flush=True))
    if res is None:
        raise SystemExit("needs Mesa (swrast_dri.so) and the reference's shaders (/root/reference or SUMA_REFERENCE_SHADERS)")
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
