#!/usr/bin/env python
"""BASELINE configs[1] verbatim: the FULL 4541-scan sequence (KITTI-00 length; 64x2048, semantic ICP, 10 GN iterations)
through the HIP pipeline and through the CPU oracle (16 threads), scan by scan: pose bits and statistics after EVERY
scan, counters (updated / new / cached surfels, submap origin) after every scan, the whole surfel buffer every
`--every` scans and at the end.  Scans are generated ahead by a process pool.  Writes one JSON line (and
gpurun_out/long_parity.json) that names the kernel sources (semantic_suma_amd/buildinfo.py) and the oracle sources it
was taken on.   usage: python tools/long_parity.py [--scans 4541] [--every 50]
Round 5: the run's 16 minutes are the oracle's, so it comes apart --  --record trace.npz (CPU only, once per oracle source)
and  --check trace.npz (GPU, one minute, the last step of tools/profile_round.sh: the evidence cannot trail the sources)."""
import argparse, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_suma_amd import synth  # noqa: E402

W, H = 2048, 64


def gen(k):
    return synth.generate_scan(k, n_azimuth=W, height=H)[:3]


def oracle_source_sha():
    """identity of the oracle sources a recorded trace belongs to (the arithmetic specification lives in them)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h")))
    for f in [os.path.join("oracle", f) for f in files] + ["include/suma_detmath.h", "include/suma_types.h"]:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def stats_row(st):
    d = st.as_dict()
    return [float(d[k]) for k in sorted(d)]


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
    ap.add_argument("--record", default=None,
                    help="CPU only: run the ORACLE alone and write its trace (pose bits, statistics and counters of every scan, "
                         "SHA-256 of the surfel buffer every --every scans) to this .npz")
    ap.add_argument("--check", default=None,
                    help="GPU only: run the HIP pipeline alone against a recorded trace -- the 16 minutes of a run are the oracle's, "
                         "so the trace is recorded once per oracle source (no GPU needed) and every build is checked in a minute")
    args = ap.parse_args()
    from semantic_suma_amd.types import params_with_size
    N = args.scans
    p = params_with_size(W, H, max_surfels=args.max_surfels)
    workers = max(2, min(32, (os.cpu_count() or 4) // 4))
    if args.record:
        return record(args, p, N, workers)
    if args.check:
        return check(args, p, N, workers)
    return live(args, p, N, workers)


def scans_ahead(pool, N, workers):
    ahead = 4 * workers
    futs = {k: pool.submit(gen, k) for k in range(min(ahead, N))}
    for k in range(N):
        sc = futs.pop(k).result()
        if k + ahead < N:
            futs[k + ahead] = pool.submit(gen, k + ahead)
        yield k, sc


def record(args, p, N, workers):
    import hashlib
    from oracle import pyoracle
    op = pyoracle.OraclePipeline(p, threads=args.threads)
    poses, stats, counts, sha_idx, shas, sizes = np.zeros((N, 4, 4)), [], np.zeros((N, 5), np.int64), [], [], []
    t0 = time.time()
    with ProcessPoolExecutor(workers) as pool:
        for k, (pts, lab, prob) in scans_ahead(pool, N, workers):
            op.process_scan(pts, lab, prob, fixed_iterations=10)
            poses[k] = op.pose()
            stats.append(stats_row(op.last_stats()))
            counts[k] = (*op.ctx.map_counts(), op.ctx.map_cached_surfels(), *op.ctx.map_submap_origin())
            if k % args.every == args.every - 1 or k == N - 1:
                sf = op.ctx.map_surfels()
                sha_idx.append(k)
                shas.append(np.frombuffer(hashlib.sha256(sf.tobytes()).digest(), np.uint8))
                sizes.append(sf.shape[0])
                print(f"scan {k}: {sf.shape[0]} surfels [{time.time() - t0:.0f} s]", file=sys.stderr, flush=True)
    np.savez_compressed(args.record, poses=poses, stats=np.array(stats), stat_keys=np.array(sorted(op.last_stats().as_dict())),
                        counts=counts, sha_idx=np.array(sha_idx), sha=np.array(shas), map_size=np.array(sizes),
                        oracle_source_sha=oracle_source_sha(), scans=N, width=W, height=H, every=args.every,
                        max_surfels=args.max_surfels, track_loss=op.track_loss(), oracle_seconds=time.time() - t0,
                        oracle_threads=args.threads)
    print(json.dumps({"recorded": args.record, "scans": N, "oracle_source_sha": oracle_source_sha(), "seconds": round(time.time() - t0, 1)}))


def check(args, p, N, workers):
    import hashlib
    from semantic_suma_amd import core
    from semantic_suma_amd.buildinfo import kernel_source_sha
    z = np.load(args.check)
    assert int(z["width"]) == W and int(z["height"]) == H and int(z["max_surfels"]) == args.max_surfels and int(z["scans"]) >= N
    hp = core.SurfelMapping(p)
    keys = [str(k) for k in z["stat_keys"]]
    sha_at = {int(k): i for i, k in enumerate(z["sha_idx"])}
    origins, max_map, compared, t_hip, t0 = set(), 0, 0, 0.0, time.time()
    with ProcessPoolExecutor(workers) as pool:
        for k, (pts, lab, prob) in scans_ahead(pool, N, workers):
            t = time.perf_counter()
            hp.processScan(pts, lab, prob, fixed_iterations=10)
            pose = hp.getCurrentPose()
            t_hip += time.perf_counter() - t
            assert np.array_equal(pose, z["poses"][k]), f"scan {k}: pose bits"
            st = hp.lastStats().as_dict()
            assert sorted(st) == keys and [float(st[q]) for q in keys] == list(z["stats"][k]), f"scan {k}: statistics"
            su, sn, cached, origin = hp.map.counts()
            assert (su, sn, cached, *origin) == tuple(int(v) for v in z["counts"][k]), f"scan {k}: counts / submap origin"
            origins.add(origin)
            if k in sha_at:
                hs = hp.map.getAllSurfels()
                assert hs.shape[0] == int(z["map_size"][sha_at[k]]), f"scan {k}: map size"
                assert hashlib.sha256(hs.tobytes()).digest() == z["sha"][sha_at[k]].tobytes(), f"scan {k}: surfel bytes"
                compared += 1
                max_map = max(max_map, hs.shape[0])
    gt = np.linalg.inv(synth.trajectory_pose(0)) @ synth.trajectory_pose(N - 1)
    drift = float(np.linalg.norm((np.linalg.inv(hp.getCurrentPose()) @ gt)[:3, 3]))
    res = {"what": "BASELINE configs[1] full sequence: HIP pipeline == CPU oracle, bit for bit (HIP run checked against the "
                   "oracle's recorded trace: pose bits, statistics, counters of every scan; SHA-256 of the whole surfel buffer)",
           "scans": N, "width": W, "height": H, "pose_and_statistics_compared": N, "surfel_buffers_compared": compared,
           "submap_origins_visited": len(origins), "max_map_surfels": max_map, "max_surfels_capacity": args.max_surfels,
           "track_loss_scans": hp.trackLoss(), "drift_m_vs_ground_truth": round(drift, 3),
           "laps_of_the_synthetic_loop": round(N * 1.1 / (2 * 450.0 + 2 * np.pi * 12.0), 2),
           "kernel_source_sha": kernel_source_sha(), "oracle_source_sha_of_the_trace": str(z["oracle_source_sha"]),
           "oracle_source_sha_here": oracle_source_sha(), "trace": os.path.relpath(args.check, ROOT),
           "oracle_seconds_when_recorded": round(float(z["oracle_seconds"]), 1), "hip_seconds": round(t_hip, 2),
           "wall_seconds": round(time.time() - t0, 1), "result": "equal"}
    assert res["oracle_source_sha_of_the_trace"] == res["oracle_source_sha_here"], "the trace was recorded on other oracle sources"
    assert res["track_loss_scans"] == int(z["track_loss"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def live(args, p, N, workers):
    from semantic_suma_amd import core
    from semantic_suma_amd.buildinfo import kernel_source_sha
    from oracle import pyoracle
    hp, op = core.SurfelMapping(p), pyoracle.OraclePipeline(p, threads=args.threads)
    t0 = time.time()
    origins, max_map, compared, t_hip, t_ora = set(), 0, 0, 0.0, 0.0
    with ProcessPoolExecutor(workers) as pool:
        for k, (pts, lab, prob) in scans_ahead(pool, N, workers):
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
           "oracle_threads": args.threads, "kernel_source_sha": kernel_source_sha(), "oracle_source_sha": oracle_source_sha(),
           "result": "equal"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
