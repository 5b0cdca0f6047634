#!/usr/bin/env python
"""Round-4 review item 7: "measure before building: whole-tile culling for the render passes".
Surfels are in creation order, so a 256-surfel tile (k_render's unit of work) is spatially coherent; a per-tile bound
could let the kernel skip tiles with no surfel in view before paying the 48-byte load and the ~600-VALU transform + facing
+ projection of phase 1a.  This script answers the question the review asked BEFORE building anything: on the
steady-state map of the bench sequence (oracle run, CPU), what fraction of the tiles has NO surfel inside
[min_depth, max_depth] x field of view at the current pose -- and, more to the point, no CANDIDATE of the render gate
(render_surfels.geom:76-103: stable, front facing, centre inside the image)?  The bar of the review: build the skip if >= 25 %.
    python tools/tile_culling_analysis.py [--scans 300] [--out profiles/r05_tile_culling_analysis.json]
Test infrastructure only (uses oracle/)."""
import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=300)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from oracle import pyoracle
    from semantic_suma_amd import synth
    from semantic_suma_amd.types import params_with_size
    W, H = 2048, 64
    p = params_with_size(W, H)
    op = pyoracle.OraclePipeline(p, threads=a.threads)
    for k in range(a.scans):
        op.process_scan(*synth.generate_scan(k, n_azimuth=W, height=H)[:3], fixed_iterations=10)
    ctx = op.ctx
    s = ctx.map_surfels()
    ts = ctx.map_timestamp()
    poses = ctx.map_poses(ts + 1).reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)  # row-major 4x4 per stamp
    inv = np.linalg.inv(op.pose().astype(np.float32).astype(np.float64))
    cr = s["count"].astype(np.int64)
    M = inv[None] @ poses[cr]                                       # inv_pose * surfelPose, render_surfels.vert:44
    pos = np.stack([s["x"], s["y"], s["z"], np.ones(len(s))], 1)
    nrm = np.stack([s["nx"], s["ny"], s["nz"], np.zeros(len(s))], 1)
    P = np.einsum("nij,nj->ni", M, pos)[:, :3]
    N = np.einsum("nij,nj->ni", M, nrm)[:, :3]
    depth = np.linalg.norm(P, axis=1)
    pitch = -np.degrees(np.arcsin(np.clip(P[:, 2] / np.maximum(depth, 1e-12), -1, 1)))
    fov_up, fov_down = abs(p.model_fov_up), abs(p.model_fov_down)
    y01 = 1.0 - (pitch + fov_up) / (fov_up + fov_down)
    in_range = (depth >= p.model_min_depth) & (depth < p.model_max_depth)
    in_fov = (y01 >= 0) & (y01 < 1)
    facing = np.einsum("ni,ni->n", N, -P / np.maximum(depth, 1e-12)[:, None]) > 0.01
    conf_thr = p.confidence_threshold
    stable = s["confidence"] > conf_thr
    recent = (cr >= ts - 100) | (s["timestamp"].astype(np.int64) >= ts - 100)
    in_view = in_range & in_fov
    cand = stable & recent & facing & in_view
    k7 = facing & in_view                                           # the fused index-map splat has no stability / age gate
    out = {"what": "fraction of k_render's 256-surfel tiles that whole-tile culling could skip, steady-state map of the bench "
                   f"sequence after {a.scans} scans (oracle, 64x2048)", "surfels": int(len(s)), "timestamp": int(ts),
           "surfel_fractions": {"in_range_and_fov": float(in_view.mean()), "stable": float(stable.mean()), "front_facing": float(facing.mean()),
                                "render_candidates": float(cand.mean()), "index_map_candidates": float(k7.mean())}}
    for T in (256, 1024):
        n = (len(s) + T - 1) // T
        pad = n * T - len(s)
        f = lambda m: np.concatenate([m, np.zeros(pad, bool)]).reshape(n, T).any(axis=1)  # noqa: E731
        out[f"tiles_of_{T}"] = {"tiles": int(n),
                                "no_surfel_in_range_and_fov": float(1.0 - f(in_view).mean()),
                                "no_render_candidate": float(1.0 - f(cand).mean()),
                                "no_index_map_candidate": float(1.0 - f(k7).mean())}
    out["verdict"] = ("below the 25 % bar of the review: tiles WITHOUT ANY surfel in view are rare because a tile is an azimuth wedge of one "
                      "scan (2 .. 75 m long) and the submap window (90 m) is about the sensor range; tiles without a render CANDIDATE already "
                      "leave k_render after the first rank barrier (the round-3 early-out), and the fused index-map pass wants every "
                      "front-facing surfel in view.  Not built.")
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
