#!/usr/bin/env python
"""BASELINE configs[1] verbatim: the FULL 4541-scan sequence (KITTI-00 length; 64x2048, semantic ICP, 10 GN iterations)
through the HIP pipeline and through the CPU oracle (16 threads), scan by scan: pose bits and statistics after EVERY
scan, counters (updated / new / cached surfels, submap origin) after every scan, the whole surfel buffer every
`--every` scans and at the end.  Scans are generated ahead by a process pool.  Writes one JSON line (and
gpurun_out/long_parity.json).   usage: python tools/long_parity.py [--scans 4541] [--every 50]"""
import argparse, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_suma_amd import synth  # noqa: E402

W, H = 2048, 64


def gen(k):
    return synth.generate_scan(k, n_azimuth=W, height=H)[:3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=4541)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--max-surfels", type=int, default=4096 * 4096,
                    help="map capacity.  The synthetic loop (975 m) is driven 5.1 times WITHOUT loop closures, both straights lie "
                         "inside one 90 m submap window, so every lap adds a drifted layer: the map passes the reference's "
                         "maxNumSurfels_ = 2048 * 2048 (SurfelMap.h:87, where the reference would silently truncate and this "
                         "library reports SUMA_ERR_CAPACITY) around scan 1500.  Default: the reference's own alternative 4096 * 4096")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "long_parity.json"))
    args = ap.parse_args()
    from semantic_suma_amd import core
    from semantic_suma_amd.types import params_with_size
    from oracle import pyoracle
    p = params_with_size(W, H, max_surfels=args.max_surfels)
    hp, op = core.SurfelMapping(p), pyoracle.OraclePipeline(p, threads=args.threads)
    t0 = time.time()
    N = args.scans
    origins, max_map, compared, t_hip, t_ora = set(), 0, 0, 0.0, 0.0
    workers = max(2, min(32, (os.cpu_count() or 4) // 4))
    with ProcessPoolExecutor(workers) as pool:
        ahead = 4 * workers
        futs = {k: pool.submit(gen, k) for k in range(min(ahead, N))}
        for k in range(N):
            pts, lab, prob = futs.pop(k).result()
            if k + ahead < N:
                futs[k + ahead] = pool.submit(gen, k + ahead)
            t = time.perf_counter()
            hp.processScan(pts, lab, prob, fixed_iterations=10)
            pose = hp.getCurrentPose()
            t_hip += time.perf_counter() - t
            t = time.perf_counter()
            op.process_scan(pts, lab, prob, fixed_iterations=10)
            t_ora += time.perf_counter() - t
            assert np.array_equal(pose, op.pose()), f"scan {k}: pose bits"
            assert hp.lastStats().as_dict() == op.last_stats().as_dict(), f"scan {k}: statistics"
            su, sn, cached, origin = hp.map.counts()
            assert (su, sn) == op.ctx.map_counts() and cached == op.ctx.map_cached_surfels(), f"scan {k}: counts"
            assert origin == op.ctx.map_submap_origin(), f"scan {k}: submap origin"
            origins.add(origin)
            if k % args.every == args.every - 1 or k == N - 1:
                hs = hp.map.getAllSurfels()
                assert hs.shape[0] == op.ctx.map_size(), f"scan {k}: map size"
                assert hs.tobytes() == op.ctx.map_surfels().tobytes(), f"scan {k}: surfel bytes"
                compared += 1
                max_map = max(max_map, hs.shape[0])
                print(f"scan {k}: {hs.shape[0]} surfels, {cached} cached, origin {origin}, x = {pose[0, 3]:.2f} y = {pose[1, 3]:.2f}  "
                      f"[{time.time() - t0:.0f} s]", file=sys.stderr, flush=True)
    gt = np.linalg.inv(synth.trajectory_pose(0)) @ synth.trajectory_pose(N - 1)
    drift = float(np.linalg.norm((np.linalg.inv(hp.getCurrentPose()) @ gt)[:3, 3]))
    res = {"what": "BASELINE configs[1] full sequence: HIP pipeline == CPU oracle, bit for bit", "scans": N, "width": W, "height": H,
           "pose_and_statistics_compared": N, "surfel_buffers_compared": compared, "submap_origins_visited": len(origins),
           "max_map_surfels": max_map, "max_surfels_capacity": args.max_surfels, "track_loss_scans": hp.trackLoss(), "drift_m_vs_ground_truth": round(drift, 3),
           "laps_of_the_synthetic_loop": round(N * 1.1 / (2 * 450.0 + 2 * np.pi * 12.0), 2),
           "oracle_seconds": round(t_ora, 1),
           "oracle_threads": args.threads, "result": "equal"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
