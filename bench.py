#!/usr/bin/env python
"""bench.py -- KITTI-style scans/s of the projective-ICP + surfel-fusion hot path on MI355X.

A step = one scan through SurfelMapping::processScan (K1-K3 preprocessing, model rendering, 10
Gauss-Newton ICP iterations + the statistics pass, map update K7-K11, post-update rendering) on a
64x2048 range image with semantic-weighted ICP (BASELINE.json configs[1]); scans are synthetic
(semantic_suma_amd/synth.py) and already resident in HBM when the timed region starts.  configs[1] is a FULL
sequence, so the timed steps come after an untimed pre-roll of the same sequence (--preroll, default 300 scans)
that brings the map to its steady size; the rate of the first scans is reported beside it as `cold_start`.
N > 1 (launched by torch.distributed.run, one process per GPU): every rank runs its own
independent sequence (sequence sharding, weak scaling) and the poses are gathered once over RCCL.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel: algorithmic bytes per launch / average launch duration, measured with
                  HIP events on the ctx stream over the timed region
  cpu_baseline -- the CPU oracle (a port of the reference path, oracle/; OpenMP over pixels / surfels, and once
                  more on one thread) timed on a bounded sample of the
                  same workload on the host cores (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# kernel label of the HIP-event scopes -> kernel symbol in rocprofv3 traces
SYMBOL = {"k6_icp_step": "k_icp_step", "k6k8_stats_radius": "k_icp_step", "k6_icp_finish": "k_icp_finish", "k4_render_surfels": "k_render",
          "k4k7_render_indexmap": "k_render", "k9_update_surfels": "k9_update", "k10_generate_surfels": "k10_generate",
          "k12_extract_submap": "k12_extract", "k7_indexmap": "k7_indexmap"}


def hbm_traffic(label, width, height):
    """HBM bytes per launch of the kernel behind `label`, from the rocprofv3 PMC passes of this very command
    (profiles/hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KB units, FETCH_SIZE
    doubled for the gfx950 wide-load undercount as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read
    from inside the process, so the figure comes from the committed passes -- and ONLY when they were taken on the
    kernel sources this run was built from (source hash recorded by tools/make_hbm_traffic.py); otherwise None."""
    try:
        from semantic_suma_amd.buildinfo import kernel_source_sha
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)
        if t.get("width") != width or t.get("height") != height or t.get("kernel_source_sha") != kernel_source_sha():
            return None
        k = t["kernels"].get(SYMBOL.get(label, label))
        return None if k is None else float(k["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def cached_scan(k, W, H):
    """synthetic scan k; with SUMA_SCAN_CACHE=<dir> the generated arrays are kept there (A/B loops inside one GPU session
    run bench.py many times on the same scans: generation is ~25 ms per scan, a load ~2 ms)"""
    from semantic_suma_amd import synth
    d = os.environ.get("SUMA_SCAN_CACHE")
    if not d:
        return synth.generate_scan(k, n_azimuth=W, height=H)[:3]
    path = os.path.join(d, f"scan_{W}x{H}_{k:06d}.npz")
    if os.path.exists(path):
        z = np.load(path)
        return z["pts"], z["lab"], z["prob"]
    pts, lab, prob = synth.generate_scan(k, n_azimuth=W, height=H)[:3]
    os.makedirs(d, exist_ok=True)
    np.savez(path + ".tmp.npz", pts=pts, lab=lab, prob=prob)
    os.replace(path + ".tmp.npz", path)
    return pts, lab, prob


def adapter_path(scans, W, H, iterations):
    """compile and run tools/adapter_bench.cpp on the first scans of the sequence (a process of its own beside this
    one: its contexts are created after the timed region is over).  None if no compiler is at hand."""
    import shutil
    import subprocess
    import tempfile
    cxx = shutil.which("g++") or shutil.which("c++") or shutil.which("hipcc")
    if cxx is None:
        return None
    tmp = tempfile.mkdtemp(prefix="suma_adapter_")
    try:
        for k, sc in enumerate(scans):
            pts, lab, prob = sc[4]
            np.ascontiguousarray(pts, dtype="<f4").tofile(os.path.join(tmp, f"{k:06d}.bin"))
            np.ascontiguousarray(lab, dtype="<f4").tofile(os.path.join(tmp, f"{k:06d}.lab"))
            np.ascontiguousarray(prob, dtype="<f4").tofile(os.path.join(tmp, f"{k:06d}.prob"))
        exe = os.path.join(tmp, "adapter_bench")
        libdir = os.path.join(ROOT, "semantic_suma_amd")
        subprocess.check_call([cxx, "-std=c++11", "-O2", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "adapter_bench.cpp"), "-o", exe, "-L", libdir, "-lsuma_hip",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], stderr=subprocess.DEVNULL)
        res = subprocess.run([exe, tmp, str(len(scans)), str(W), str(H), str(iterations)], capture_output=True,
                             timeout=240)
        if res.returncode != 0:
            print(f"adapter_bench failed ({res.returncode}): {res.stderr.decode()[-300:]}", file=sys.stderr)
            return None
        return json.loads(res.stdout.decode().strip().splitlines()[-1])
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        print(f"adapter_bench not run: {e!r}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_other_mode(args, rank, local_rank, world, coll_dev):
    """BASELINE configs[2] (8 hypotheses per scan sharded over the ranks, one gather per scan) and configs[3] (the 11
    KITTI odometry sequences, LPT-assigned to the ranks, several pipelines per GPU where a rank owns several)."""
    import torch
    import torch.distributed as dist
    from semantic_suma_amd import core, synth
    from semantic_suma_amd.distributed import gather_poses, lpt_assign, run_hypotheses_hip, run_sequences_hip
    from semantic_suma_amd.types import params_with_size
    W, H, K, Wu = args.width, args.height, args.steps, args.warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    up = core.Context(params_with_size(W, H, max_surfels=65536), device=local_rank)  # only to place scans in HBM

    def resident(sc):
        return tuple(up.device_array(x) for x in sc) + (sc[0].shape[0],)

    if args.mode == "hypotheses":
        n_hyp = 8
        p = params_with_size(W, H, max_iterations=args.icp_iterations, stopping_threshold=0.0, delta=0.0)
        scans = [resident(synth.generate_scan(k, n_azimuth=W, height=H)[:3]) for k in range(Wu + K)]
        gather = lambda a: gather_poses(a, device=coll_dev)  # noqa: E731
        # warm-up: the process (module load, first-touch allocations), not the map -- the timed run starts a fresh one
        run_hypotheses_hip(p, scans[:max(2, Wu)], n_hyp, rank, world, device=local_rank, gather=gather, on_device=True)
        barrier()
        t0 = time.perf_counter()
        poses, winners = run_hypotheses_hip(p, scans, n_hyp, rank, world, device=local_rank, gather=gather, on_device=True)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            n = len(scans)
            print(json.dumps({"metric": "scans_per_sec", "value": n / elapsed, "unit": "scans/s", "n_gpus": world, "steps": n,
                              "warmup": Wu, "ms_per_step": 1000.0 * elapsed / n, "higher_is_better": True, "scaling": "strong",
                              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "hypotheses_per_sec": n_hyp * (n - 1) / elapsed,
                              "config": {"workload": f"BASELINE configs[2]: {H}x{W}, {n_hyp} ICP hypotheses per scan "
                                                     f"({args.icp_iterations} GN iterations each) sharded over the ranks, one "
                                                     "exchange of 18 doubles per hypothesis and scan, map update with the winner "
                                                     "on every rank; native host loop (suma_run_hypotheses), scans resident in HBM",
                                         "winners": winners[1:9], "pose_x_end": round(float(poses[-1][0, 3]), 3),
                                         "parallelism": f"hypothesis-sharded x{world}"}}))
        return
    # sequences11
    lengths_full = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]  # KITTI odometry 00-10
    scale = max(1, int(os.environ.get("SUMA_SEQ_SCALE", "32")))
    conc = max(1, int(os.environ.get("SUMA_SEQ_CONCURRENT", "4")))
    lengths = [max(4, n // scale) for n in lengths_full]
    assign, loads = lpt_assign(lengths, world)
    p = params_with_size(W, H)
    mine = assign[rank]
    cache = {s: [resident(synth.generate_scan(1337 * (s + 1) + k, n_azimuth=W, height=H)[:3]) for k in range(lengths[s])]
             for s in mine}
    run_sequences_hip(p, {-1: cache[mine[0]][:3]} if mine else {}, device=local_rank, fixed_iterations=args.icp_iterations,
                      max_concurrent=1, on_device=True)  # warm the process
    barrier()
    t0 = time.perf_counter()
    res = run_sequences_hip(p, cache, device=local_rank, fixed_iterations=args.icp_iterations, max_concurrent=conc,
                            on_device=True)
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    ends = np.zeros((len(lengths), 16))
    for s, (n, pose) in res.items():
        ends[s] = pose.ravel()
    allp = gather_poses(ends, device=coll_dev) if world > 1 else ends[None]  # the one collective: trajectory end poses
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    busy = gather_poses(np.array([mine_s]), device=coll_dev)[:, 0] if world > 1 else np.array([mine_s])
    if rank == 0:
        total = sum(lengths)
        print(json.dumps({"metric": "scans_per_sec", "value": total / elapsed, "unit": "scans/s", "n_gpus": world, "steps": total,
                          "warmup": 0, "ms_per_step": 1000.0 * elapsed / total, "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"BASELINE configs[3]: the 11 KITTI odometry sequence lengths / {scale} "
                                                 f"({lengths}), {H}x{W}, semantic ICP ({args.icp_iterations} GN iterations), "
                                                 f"LPT-assigned to the ranks, up to {conc} concurrent pipelines per GPU; native "
                                                 "host loop (suma_run_sequences), scans resident in HBM",
                                     "assignment": assign, "load_scans": loads,
                                     "idle_frac_per_rank": [round(1.0 - float(b) / elapsed, 3) for b in busy],
                                     "sequences_done": int((np.abs(allp).sum(axis=(0, 2)) > 0).sum()),
                                     "parallelism": f"sequence-sharded x{world} (LPT)"}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--icp-iterations", type=int, default=10)
    ap.add_argument("--cpu-scans", type=int, default=20, help="scans of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the untimed per-kernel HIP-event pass")
    ap.add_argument("--profile-scans", type=int, default=20, help="scans of the untimed per-kernel pass")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--preroll", type=int, default=300,
                    help="scans of the same sequence processed (untimed) before the timed region, so that the map has its "
                         "steady size; 0 = time the sequence from its first scan (e.g. --steps 4541 --preroll 0)")
    ap.add_argument("--max-surfels", type=int, default=0,
                    help="map capacity (0 = the reference's maxNumSurfels_ = 2048 * 2048, SurfelMap.h:87; the synthetic loop "
                         "is driven several times without loop closures, so a full 4541-scan run layers the map and needs "
                         "the reference's own alternative 4096 * 4096)")
    ap.add_argument("--adapter-scans", type=int, default=120,
                    help="scans of the class-by-class adapter path timing (tools/adapter_bench.cpp; 0 = skip)")
    ap.add_argument("--mode", default="single", choices=["single", "hypotheses", "sequences11", "adapter"],
                    help="single: BASELINE configs[1] (the bench contract); hypotheses: configs[2], 8 ICP hypotheses per scan "
                         "sharded over the ranks; sequences11: configs[3], the 11 KITTI sequence lengths (scaled) LPT-assigned")
    ap.add_argument("--kernels-json", default=os.path.join(ROOT, "gpurun_out", "bench_kernels.json"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one process per GPU; SUMA_BENCH_FORCE_DEVICE is a debugging aid (several ranks on one GPU with gloo)
    local_rank = int(os.environ.get("SUMA_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")

    from semantic_suma_amd import core, synth
    from semantic_suma_amd.distributed import gather_poses
    from semantic_suma_amd.types import params_with_size

    W, H, K, Wu = args.width, args.height, args.steps, args.warmup
    if args.mode == "adapter":  # only the host-language comparison of tools/adapter_bench.cpp (no torch.distributed)
        n = max(10, args.adapter_scans)
        print(json.dumps(adapter_path([(None, None, None, 0, cached_scan(k, W, H)) for k in range(n)], W, H,
                                      args.icp_iterations)))
        return
    if args.mode != "single":
        run_other_mode(args, rank, local_rank, world, coll_dev)
        if world > 1:
            dist.destroy_process_group()
        return
    kitti_dir = os.environ.get("SUMA_KITTI_DIR")  # e.g. .../sequences/00 ; absent on the build / bench machines
    # the reference fetches label i+4 / prob i+5 for point i (Preprocessing.cpp:142-145, an offset bug that is harmless
    # on its zero-padded network output); reproduced for parity on synthetic data, switched off for real labels
    extra = dict(max_surfels=args.max_surfels) if args.max_surfels else {}
    p = params_with_size(W, H, label_offset=0, prob_offset=0, **extra) if kitti_dir else params_with_size(W, H, **extra)
    pipe = core.SurfelMapping(p, device=local_rank)
    ctx = pipe.ctx

    # ---- synthetic sequence of this rank (its own stretch of the trajectory), uploaded to HBM
    # Order of a run: Wu warm-up scans | K scans timed from the cold start (extra key `cold_start`) | untimed pre-roll
    # until `--preroll` scans have gone through (the map reaches its steady size: the active area is bounded by the
    # submap window, SurfelMap.cpp:667-677) | the K TIMED scans of the contract (`value`) | E scans with every kernel
    # group bracketed.  BASELINE configs[1] is a FULL sequence: its rate is the steady-state rate, not the rate of
    # the first scans on a near-empty map (round-2 review).  --preroll 0 times the sequence from its first scan.
    k0 = 400 * rank
    scans = []
    t_gen = time.perf_counter()
    E = 0 if args.no_kernel_events else max(0, min(args.profile_scans, K))
    PR = max(0, args.preroll)
    cold = PR >= Wu + K + 20          # room for a separate cold-start measurement in front of the pre-roll
    n_before = max(PR, Wu)            # scans processed before the timed region starts
    total = n_before + K + E
    seq = None
    if kitti_dir:
        from semantic_suma_amd import kitti
        seq = kitti.Sequence(kitti_dir)
    for k in range(total):
        if seq is not None:
            pts, lab, prob = seq[(k0 + k) % len(seq)]
        else:
            pts, lab, prob = cached_scan(k0 + k, W, H)
        scans.append((ctx.device_array(pts), ctx.device_array(lab), ctx.device_array(prob), pts.shape[0],
                      (pts, lab, prob) if (rank == 0 and k < max(args.cpu_scans, args.adapter_scans)) else None))
    t_gen = time.perf_counter() - t_gen
    n_points = float(np.mean([s[3] for s in scans]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def run(a, b):
        for k in range(a, b):
            pipe.processScanDevice(*scans[k][:4], fixed_iterations=args.icp_iterations)

    run(0, Wu)
    cold_start = None
    done = Wu
    if cold:
        ctx.synchronize()
        tc = time.perf_counter()
        run(Wu, Wu + K)
        ctx.synchronize()
        tc = time.perf_counter() - tc
        cold_start = {"value": K / tc, "unit": "scans/s", "ms_per_step": 1000.0 * tc / K, "scans": K, "after_scans": Wu,
                      "map_surfels_end": pipe.map.size()}
        done = Wu + K
    run(done, n_before)
    map_size_start = pipe.map.size()
    # timed region: only the Gauss-Newton chain (the dominant kernel) is bracketed by HIP events -- one event
    # pair per chain of identical launches, on every 4th scan (an event record costs the stream a ~6 us bubble)
    ctx.profile(3)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    run(n_before, n_before + K)
    poses = gather_poses(pipe.getCurrentPose(), device=coll_dev) if world > 1 else None
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert poses.shape[0] == world

    dominant = [k for k in ctx.profile_get() if k["launches"]]
    map_size = pipe.map.size()
    pose = pipe.getCurrentPose()
    # untimed continuation of the same sequence with every kernel group bracketed: the per-kernel table
    kernels = []
    if E:
        ctx.profile(1)
        ctx.profile_reset()
        run(n_before + K, n_before + K + E)
        kernels = ctx.profile_get()
    ctx.profile(0)
    if seq is None:  # synthetic trajectory: known ground truth
        gt = np.linalg.inv(synth.trajectory_pose(k0)) @ synth.trajectory_pose(k0 + n_before + K - 1)
        drift = float(np.linalg.norm((np.linalg.inv(pose) @ gt)[:3, 3]))
    else:
        drift = float("nan")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "scans_per_sec", "value": world * K / elapsed, "unit": "scans/s", "n_gpus": world, "steps": K,
        "warmup": Wu, "ms_per_step": 1000.0 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "kitti" if seq is not None else "synthetic",
        "config": {"workload": f"BASELINE configs[1]: synthetic KITTI-like sequence, {H}x{W} range images, "
                               f"semantic-weighted ICP ({args.icp_iterations} GN iterations + stats pass) + surfel "
                               "fusion, one sequence per GPU, scans resident in HBM; timed after "
                               f"{n_before} untimed scans of the same sequence ({Wu} warm-up + pre-roll to the steady "
                               f"map size: {map_size_start} surfels at the start of the timed region)",
                   "points_per_scan": round(n_points), "preroll_scans": n_before, "map_surfels_start": map_size_start,
                   "map_surfels_end": map_size, "drift_m": round(drift, 4),
                   "parallelism": f"sequence-sharded x{world}" if world > 1 else "single GPU"},
    }
    if cold_start is not None:
        out["cold_start"] = cold_start  # the first K scans of the sequence (growing map): NOT the headline

    def derive(ks):
        for k in ks:
            k["avg_us"] = 1000.0 * k["total_ms"] / max(k["launches"], 1)
            k["bytes_per_launch"] = k["bytes"] / max(k["launches"], 1)
            k["gbps"] = k["bytes"] / max(k["total_ms"], 1e-9) / 1e6
        ks.sort(key=lambda k: -k["total_ms"])
        return ks

    derive(kernels)
    if derive(dominant):
        dom = dominant[0]  # measured over the TIMED region
        out["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": dom["gbps"], "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": dom["gbps"] / HBM_PEAK_GBPS,
                           "traffic": hbm_traffic(dom["name"], W, H),
                           "avg_launch_us": dom["avg_us"], "bytes_per_launch": dom["bytes_per_launch"],
                           "launches": dom["launches"],
                           "launches_in_timed_region": K * args.icp_iterations,
                           "share_of_timed_region": dom["avg_us"] * K * args.icp_iterations / (1e6 * elapsed)}
        if kernels:
            out["roofline"]["kernel_time_share"] = next(
                (k["total_ms"] for k in kernels if k["name"] == dom["name"]), 0.0) / sum(k["total_ms"] for k in kernels)
    if kernels:
        try:
            os.makedirs(os.path.dirname(args.kernels_json), exist_ok=True)
            with open(args.kernels_json, "w") as f:
                json.dump({"elapsed_s": elapsed, "steps": K, "profiled_scans": E, "timed_region_dominant": dominant,
                           "kernels": kernels}, f, indent=1)
        except OSError:
            pass
        print("kernel".ljust(26) + "launches  avg_us   GB/s(alg)  share", file=sys.stderr)
        tot = sum(k["total_ms"] for k in kernels)
        for k in kernels:
            print(f"{k['name']:<26}{k['launches']:>8}{k['avg_us']:>9.1f}{k['gbps']:>11.1f}{100 * k['total_ms'] / tot:>7.1f}%",
                  file=sys.stderr)
        print(f"(untimed pass of {E} scans, every kernel group bracketed: {tot:.1f} ms of kernel time); "
              f"timed region {1000 * elapsed:.1f} ms for {K} scans; scan generation {t_gen:.1f} s", file=sys.stderr)

    # ---- CPU baseline: the oracle (port of the reference path) on the first scans of the same sequence, built
    #      -O3 -march=native on this very host (bit-identical to the -O2 build the tests use:
    #      tests/test_oracle_kat.py::test_native_build_is_bit_identical)
    if world == 1 and args.cpu_scans > 0:
        from oracle import pyoracle
        variant = "native" if pyoracle.build_native() else ""
        n_cpu = min(args.cpu_scans, total)

        def time_oracle(threads):
            op = pyoracle.OraclePipeline(p, variant=variant, threads=threads)
            tc = time.perf_counter()
            for k in range(n_cpu):
                pts, lab, prob = scans[k][4]
                op.process_scan(pts, lab, prob, fixed_iterations=args.icp_iterations)
            return n_cpu / (time.perf_counter() - tc)

        # (a) the sequential restatement, (b) the same code with OpenMP over pixels / surfels (bit-identical
        # results: integer sums, z-buffer minima and stable compactions are order independent)
        ncores = os.cpu_count() or 1
        threads = max(1, min(ncores, 16))  # tools/oracle_scaling.py on the 256-core MI355X host: 8 -> 29, 16 -> 48, 32 -> 36, 64 -> 24 scans/s
        flags = "-O3 -march=native" if variant else "-O2"
        sample = f"first {n_cpu} scans of the same {H}x{W} sequence through oracle/ (gcc {flags})"
        out["cpu_baseline"] = {"value": time_oracle(threads), "unit": "scans/s", "cores": threads, "kind": "port",
                               "sample": f"{sample}, OpenMP, {threads} threads of {ncores} host cores"}
        out["cpu_baseline_single_thread"] = {"value": time_oracle(1), "unit": "scans/s", "cores": 1, "kind": "port",
                                             "sample": f"{sample}, one thread"}
    # ---- what a host pays that does NOT use the scan pipeline: the class-by-class call sequence of the reference's
    #      processScan on the adapter classes (include/suma_adapter.hpp), the phase API with (empty) loop-closure hooks,
    #      and the one-call pipeline -- one C++ driver, the same host vectors (tools/adapter_bench.cpp)
    if world == 1 and args.adapter_scans > 0:
        ap_res = adapter_path(scans[:min(args.adapter_scans, total)], W, H, args.icp_iterations)
        if ap_res is not None:
            out["adapter_path"] = ap_res
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
