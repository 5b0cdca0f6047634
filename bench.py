#!/usr/bin/env python
"""bench.py -- KITTI-style scans/s of the projective-ICP + surfel-fusion hot path on MI355X.

A step = one scan through SurfelMapping::processScan (K1-K3 preprocessing, model rendering, 10
Gauss-Newton ICP iterations + the statistics pass, map update K7-K11, post-update rendering) on a
64x2048 range image with semantic-weighted ICP (BASELINE.json configs[1]); scans are synthetic
(semantic_suma_amd/synth.py) and already resident in HBM when the timed region starts.  configs[1] is a FULL
sequence, so the timed steps come after an untimed pre-roll of the same sequence (--preroll, default 300 scans)
that brings the map to its steady size; the rate of the first scans is reported beside it as `cold_start`.
N > 1: one process per GPU under torch.distributed.run -- started by the driver, or by bench.py itself when
`python bench.py --gpus N` is run without a launcher (no WORLD_SIZE in the environment: the process re-executes
itself through `python -m torch.distributed.run --nproc-per-node N`, see self_launch_if_needed).  Every rank runs
its own independent sequence (sequence sharding, weak scaling) and the poses are gathered once over RCCL; the JSON
line carries `rccl_ranks` (the process group's world size), `backend` and the device index of every rank.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel: algorithmic bytes per launch / average launch duration, measured with
                  HIP events on the ctx stream over the timed region
  cpu_baseline -- the CPU oracle (a port of the reference path, oracle/; OpenMP over pixels / surfels, and once
                  more on one thread) timed on a bounded sample of the
                  same workload on the host cores (rank 0, N = 1 only)
  cpu_baseline_reference_gl (optional, only where Mesa and the reference's shader sources exist) -- the reference's own
                  GLSL executed by llvmpipe on the host CPU: the literal "reference path on the CPU" (kind "reference")
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


def usable_cpus():
    """CPUs this process may really use: the cgroup quota (cpu.max) where there is one, else the visible cores"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))  # taskset / cpuset
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def _gen_job(args):
    k, W, H = args
    return k, cached_scan(k, W, H)


def generate_scans(ks, W, H):
    """synthetic scans for the indices ks (in order); long runs (the literal 4541-scan sequence) are generated by a pool
    of worker processes -- one scan is ~25 ms of single-threaded numpy"""
    ks = list(ks)
    # every rank of an N-GPU run generates its own stretch at the same time: the host's CPUs (what the cgroup grants, not
    # what the box shows -- the pool's boxes show 256 cores and grant 16) are shared between the ranks' pools
    workers = min(32, max(1, usable_cpus() // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    if len(ks) < 400 or workers < 2:
        for k in ks:
            yield cached_scan(k, W, H)
        return
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as pool:
        for k, sc in pool.map(_gen_job, [(k, W, H) for k in ks], chunksize=8):
            yield sc


def adapter_path(scans, W, H, iterations, preroll=0):
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
        res = subprocess.run([exe, tmp, str(len(scans)), str(W), str(H), str(iterations), str(preroll)],
                             capture_output=True, timeout=600)
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
                              "hypotheses_per_sec": n_hyp * (n - 1) / elapsed, **args.dist_info,
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
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", **args.dist_info,
                          "config": {"workload": f"BASELINE configs[3]: the 11 KITTI odometry sequence lengths / {scale} "
                                                 f"({lengths}), {H}x{W}, semantic ICP ({args.icp_iterations} GN iterations), "
                                                 f"LPT-assigned to the ranks, up to {conc} concurrent pipelines per GPU; native "
                                                 "host loop (suma_run_sequences), scans resident in HBM",
                                     "assignment": assign, "load_scans": loads,
                                     "idle_frac_per_rank": [round(1.0 - float(b) / elapsed, 3) for b in busy],
                                     "sequences_done": int((np.abs(allp).sum(axis=(0, 2)) > 0).sum()),
                                     "parallelism": f"sequence-sharded x{world} (LPT)"}}))


def _all_devices(dist, world, local_rank, coll_dev):
    """the HIP device index of every rank (all-gathered: one per rank unless SUMA_BENCH_FORCE_DEVICE shares a GPU)"""
    if world == 1:
        return [local_rank]
    import torch
    t = torch.tensor([local_rank], dtype=torch.int64, device=coll_dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [int(o.item()) for o in outs]


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def relaunch_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N` (N > 1) re-executes itself as when no launcher has set WORLD_SIZE: one
    process per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container hostname may not
    resolve).  argv = this process's own arguments (sys.argv[1:]), passed through unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port()),
            os.path.abspath(__file__)] + list(argv)


def self_launch_if_needed(args, argv=None, environ=None, run=None):
    """`python bench.py --gpus 8` without a launcher used to run ONE rank (round-3 review): now the process becomes the
    launcher.  Returns None when this process is a rank (or N == 1), else the launcher's exit code."""
    environ = os.environ if environ is None else environ
    if args.gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    import subprocess
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this platform
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = relaunch_command(args.gpus, sys.argv[1:] if argv is None else argv)
    print("bench.py: launching " + " ".join(cmd[1:6]) + " ...", file=sys.stderr)
    return (run or subprocess.call)(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--icp-iterations", type=int, default=10)
    ap.add_argument("--cpu-scans", type=int, default=20, help="scans of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--reference-gl-scans", type=int, default=3,
                    help="scans of the optional cpu_baseline_reference_gl leg (the reference's GLSL in Mesa llvmpipe; runs only "
                         "where the shaders and Mesa exist; 0 = skip)")
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
                    help="scans of the class-by-class adapter path timing (tools/adapter_bench.cpp; 0 = skip), taken at the "
                         "start of the timed region, i.e. after the same pre-roll (capped at 2 x --steps)")
    ap.add_argument("--host-vector-scans", type=int, default=100,
                    help="least number of scans of the host-vector entry's sample (it is max(--steps, this))")
    ap.add_argument("--no-host-vectors", action="store_true",
                    help="skip the host-vector entry (suma_pipeline_process_scan from pageable arrays) timed behind the contract's region")
    ap.add_argument("--no-reference-mode", action="store_true",
                    help="skip the extra stretch in the reference's own Gauss-Newton mode (stopping tests, max iterations 33)")
    ap.add_argument("--no-loop-closure", action="store_true", help="skip the loop-closure verification timing (batched vs serial)")
    ap.add_argument("--mode", default="single", choices=["single", "hypotheses", "sequences11", "adapter"],
                    help="single: BASELINE configs[1] (the bench contract); hypotheses: configs[2], 8 ICP hypotheses per scan "
                         "sharded over the ranks; sequences11: configs[3], the 11 KITTI sequence lengths (scaled) LPT-assigned")
    ap.add_argument("--kernels-json", default=os.path.join(ROOT, "gpurun_out", "bench_kernels.json"))
    args = ap.parse_args()
    rc = self_launch_if_needed(args)
    if rc is not None:
        raise SystemExit(rc)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):  # a scaling point measured on another number of ranks than it claims is worthless
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
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
    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    # what the collective layer really saw (the round-3 review asked for proof that N ranks met over RCCL)
    args.dist_info = {"rccl_ranks": dist.get_world_size() if world > 1 else 1,
                      "backend": (dist.get_backend() if world > 1 else None),
                      "devices": sorted(set(_all_devices(dist, world, local_rank, coll_dev)))}

    from semantic_suma_amd import core, synth
    from semantic_suma_amd.distributed import NativeGather, gather_poses
    from semantic_suma_amd.types import params_with_size

    W, H, K, Wu = args.width, args.height, args.steps, args.warmup
    if args.mode == "adapter":  # only the host-language comparison of tools/adapter_bench.cpp (no torch.distributed)
        n = max(10, args.adapter_scans)
        pr = max(0, args.preroll)
        print(json.dumps(adapter_path([(None, None, None, 0, sc) for sc in generate_scans(range(pr + n), W, H)], W, H,
                                      args.icp_iterations, preroll=pr)))
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
    # ONE collective path for the data: the pose gather of an N-GPU run goes through the C-ABI export of SURVEY.md 8(b)
    # (suma_gather_poses, libsuma_hip_dist.so: an RCCL all-gather on the ctx stream); torch.distributed bootstraps it and
    # carries the launcher's barriers and the max-over-ranks clock.  gloo runs (CPU tests, several ranks on one device:
    # RCCL refuses that) and a rank that cannot create its communicator use torch.distributed, and the line says so.
    native = NativeGather(ctx, device=coll_dev) if (world > 1 and args.backend == "nccl") else None
    if native is not None and not native.ok:
        print(f"bench.py: native pose gather unavailable ({native.error}); using torch.distributed", file=sys.stderr)
    args.dist_info["pose_gather"] = ("libsuma_hip_dist.so: suma_gather_poses (RCCL all-gather on the ctx stream)"
                                     if native is not None and native.ok else
                                     ("torch.distributed all_gather (" + (native.error if native is not None else
                                      f"backend {args.backend}") + ")") if world > 1 else None)

    def gather_pose(a):
        return native.gather(a) if native is not None and native.ok else gather_poses(a, device=coll_dev)

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
    # scans of the host-vector entry, timed right behind the contract's region: at least 100 (the driver runs --steps 20:
    # a 20-scan sample is 7 ms, one host hiccup of a few ms moves it by a third -- round-4 review, weak point 3)
    HV = 0 if args.no_host_vectors else max(K, args.host_vector_scans)
    # ... behind HVW untimed host-vector calls: the stretch follows a stretch of RESIDENT scans, during which the copy
    # stream and the staging helpers sit idle, and the first host-vector call after >= 10 ms of that was measured at
    # 3 - 6 ms twice (profiles/r05_host_vector_calls.txt); a caller that feeds host vectors scan after scan -- the thing
    # this key measures -- never has that gap
    HVW = 3 if HV else 0
    # ... and RM scans in the reference's OWN Gauss-Newton mode at the very end: stopping tests on, `max iterations` of the
    # parameter block (default.xml: 33) -- LieGaussNewton.cpp:23-33, what an unchanged caller gets with fixed_iterations = 0
    RM = 0 if args.no_reference_mode else 40
    total = n_before + K + HVW + HV + E + RM
    seq = None
    if kitti_dir:
        from semantic_suma_amd import kitti
        seq = kitti.Sequence(kitti_dir)
    source = (seq[(k0 + k) % len(seq)] for k in range(total)) if seq is not None else generate_scans(range(k0, k0 + total), W, H)
    for k, (pts, lab, prob) in enumerate(source):
        # host arrays are kept where a host-side consumer needs them: the host-vector entry (every rank), and on rank 0
        # the CPU baseline and the adapter path, which replay the sequence from its first scan up to their sample
        keep = (n_before - 3 <= k < n_before) or (n_before + K <= k < n_before + K + HVW + HV) or \
               (rank == 0 and world == 1 and (args.cpu_scans > 0 or args.adapter_scans > 0) and k < n_before + K + HVW + HV)
        scans.append((ctx.device_array(pts), ctx.device_array(lab), ctx.device_array(prob), pts.shape[0],
                      (pts, lab, prob) if keep else None))
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

    # The timed regions are driven by the NATIVE loop over processScan (suma_pipeline_run_scans, include/suma_runner.h) --
    # the reference's caller is a C++ loop too (VisualizerWindow.cpp:636-689 -> SurfelMapping.cpp:175); an interpreter
    # between two scans costs 2-3 % of a 0.32 ms scan.  The jobs are marshalled before the clock starts.
    timed_job = pipe.prepareScans([scans[k][:4] for k in range(n_before, n_before + K)], True)
    host_warm = pipe.prepareScans([scans[k][4] for k in range(n_before + K, n_before + K + HVW)], False) if HV else None
    host_job = pipe.prepareScans([scans[k][4] for k in range(n_before + K + HVW, n_before + K + HVW + HV)], False) if HV else None

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
    run(done, max(done, n_before - 3))
    for k in range(max(done, n_before - 3), n_before):  # the last pre-roll scans go through the host-vector entry: its
        if HV:                                           # staging buffers and copy threads exist before it is timed
            pipe.processScan(*scans[k][4], fixed_iterations=args.icp_iterations)
        else:
            run(k, k + 1)
    map_size_start = pipe.map.size()
    # timed region: only the Gauss-Newton chain (the dominant kernel) is bracketed by HIP events -- one event
    # pair per chain of identical launches, on every 4th scan (an event record costs the stream a ~6 us bubble)
    ctx.profile(3)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    timed_call_s = np.zeros(K, dtype=np.float64)  # host time of every call of the timed region (filled by the native loop)
    assert pipe.runScans(timed_job, True, fixed_iterations=args.icp_iterations, call_seconds=timed_call_s) == K
    ctx.synchronize()
    own = time.perf_counter() - t0  # this rank's scans alone: everything behind it is waiting for the slowest rank
    poses = gather_pose(pipe.getCurrentPose()) if world > 1 else None
    if native is not None:  # the one data-path collective of the run is done: every rank destroys its communicator here
        native.close()
        native = None
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert poses.shape[0] == world
        # one 8-GPU run should carry its whole diagnosis: every rank's own rate and the share of the timed region it
        # spent waiting for the others (gather + barrier)
        owns = gather_poses(np.array([own]), device=coll_dev)[:, 0]
        per_rank = {"scans_per_sec": [round(K / float(o), 1) for o in owns],
                    "idle_frac": [round(1.0 - float(o) / elapsed, 4) for o in owns]}

    dominant = [k for k in ctx.profile_get() if k["launches"]]
    map_size = pipe.map.size()
    pose = pipe.getCurrentPose()
    # ---- the same pipeline fed the way the reference's caller feeds processScan(const rv::Laserscan&)
    #      (SurfelMapping.cpp:175, 323-331): pageable host vectors, one blocking call per scan, no look-ahead.  The next K
    #      scans of the sequence; NOT the headline (the bench contract times resident inputs), reported beside it.
    host_vectors = None
    if HV:
        ctx.profile(0)
        warm_s = np.zeros(HVW, dtype=np.float64)
        pipe.runScans(host_warm, False, fixed_iterations=args.icp_iterations, call_seconds=warm_s)  # untimed, see HVW
        barrier()
        call_s = np.zeros(HV, dtype=np.float64)  # host time of each blocking call: shows a stall as what it is, one long call
        pipe.hostEntryTimes(reset=True)
        th = time.perf_counter()
        assert pipe.runScans(host_job, False, fixed_iterations=args.icp_iterations, call_seconds=call_s) == HV
        barrier()
        th = time.perf_counter() - th
        per_call = call_s * 1e6
        breakdown = pipe.hostEntryTimes()
        if world > 1:
            t = torch.tensor([th], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            th = float(t.item())
        host_vectors = {"value": world * HV / th, "unit": "scans/s", "ms_per_step": 1000.0 * th / HV, "scans": HV,
                        "vs_resident": (HV / th) / (K / elapsed),
                        "call_us": {"median": round(float(np.median(per_call)), 1), "max": round(float(per_call.max()), 1),
                                    "calls_over_twice_the_median": int((per_call > 2.0 * np.median(per_call)).sum()),
                                    "first_30": [round(float(x), 1) for x in per_call[:30]],
                                    "untimed_warm_up_calls": [round(float(x) * 1e6, 1) for x in warm_s]},
                        # where the caller's thread spent a call, averaged (us): waiting for the staging slot, copying
                        # pageable -> pinned (copy_threads threads), enqueueing the upload, enqueueing kernels, and
                        # waiting for the minimisation result (= the GPU is the bottleneck, as with resident scans)
                        "call_breakdown_us": breakdown,
                        "cpus": {"usable": usable_cpus(), "affinity": len(os.sched_getaffinity(0)), "visible": os.cpu_count()},
                        "entry": "suma_pipeline_process_scan: pageable host vectors -> pinned staging (8 cores) -> copy "
                                 "stream -> K1-K3 on the side stream; the PCIe-inclusive rate, same pipeline, the next "
                                 f"{HV} scans of the sequence, driven by the same native loop as the timed region"}
    # untimed continuation of the same sequence with every kernel group bracketed: the per-kernel table
    kernels = []
    if E:
        ctx.profile(1)
        ctx.profile_reset()
        run(n_before + K + HVW + HV, n_before + K + HVW + HV + E)
        kernels = ctx.profile_get()
    ctx.profile(0)
    # ---- the device side of a loop-closure verification (SurfelMapping.cpp:679-757: render_inactive, three
    #      initial guesses minimised + evaluated, render_composed + evaluation for a guess that passes) on the steady map,
    #      batched (suma_loop_closure_verify: the guesses are ONE Gauss-Newton chain, grid.y = guess) against the reference's
    #      one-by-one sequencing (suma_loop_closure_verify_serial) and against a single guess.  Untimed extra, rank 0.
    loop_closure = None
    if rank == 0 and world == 1 and not args.no_loop_closure:
        try:
            cur = pipe.frame(0)
            prior = pipe.getCurrentPose().astype(np.float64)
            Rz = np.eye(4)
            Rz[:2, :2] = -Rz[:2, :2]
            half = np.eye(4)
            half[0, 3] = 0.5
            inits = [np.eye(4), Rz, half]
            ctv = float(p.confidence_threshold)
            reps = 20

            def clock(guesses, gates, serial):
                core.loop_closure_verify(ctx, cur, prior, guesses, prior, ctv, *gates, serial=serial)  # warm
                ctx.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    r = core.loop_closure_verify(ctx, cur, prior, guesses, prior, ctv, *gates, serial=serial)
                ctx.synchronize()
                return 1e6 * (time.perf_counter() - t) / reps, [bool(x["passed"]) for x in r], [int(x["after_minimize"]["iterations"]) for x in r]

            loop_closure = {"n_init": 3, "repetitions": reps, "map_surfels": pipe.map.size(), "unit": "us per verification"}
            for name, gates in (("reference_gates", (0.2, 0.85)), ("no_guess_passes", (2.0, 0.85))):
                b, pb, it = clock(inits, gates, False)
                sq, ps, _ = clock(inits, gates, True)
                one, _, _ = clock(inits[:1], gates, False)
                assert pb == ps
                # gn_iterations: steps each guess's chain took before the stopping tests fired (max iterations = the chain
                # never converged and all its launches worked: then three chains side by side cost what the batched pixel
                # phase costs, ~2 x one chain; chains that converge early are where batching pays: tools/loop_closure_timing.py)
                loop_closure[name] = {"batched": round(b, 1), "serial": round(sq, 1), "one_guess": round(one, 1),
                                      "batched_over_one_guess": round(b / one, 3), "serial_over_batched": round(sq / b, 3),
                                      "passed": pb, "gn_iterations": it, "max_iterations": int(p.max_iterations)}
        except Exception as e:  # noqa: BLE001 -- an extra must never cost the bench line
            print(f"loop_closure leg not taken: {e!r}", file=sys.stderr)
    reference_mode = None
    if RM:
        first = n_before + K + HVW + HV + E
        job = pipe.prepareScans([scans[k][:4] for k in range(first, first + RM)], True)
        pipe.processScanDevice(*scans[first][:4], fixed_iterations=0)  # untimed: the mode's first scan
        job = pipe.prepareScans([scans[k][:4] for k in range(first + 1, first + RM)], True)
        barrier()
        tr = time.perf_counter()
        assert pipe.runScans(job, True, fixed_iterations=0) == RM - 1
        ctx.synchronize()
        tr = time.perf_counter() - tr
        st = pipe.minimizeStats()
        reference_mode = {"value": (RM - 1) / tr, "unit": "scans/s", "scans": RM - 1, "max_iterations": int(p.max_iterations),
                          "stopping_threshold": float(p.stopping_threshold), "delta": float(p.delta),
                          "last_scan_gn_iterations": int(st.iterations), "last_scan_converged": bool(st.converged),
                          "map_surfels": pipe.map.size(),
                          "what": "the same pipeline with fixed_iterations = 0: LieGaussNewton's stopping tests decide "
                                  "(LieGaussNewton.cpp:23-33, 64-66), at most `max iterations` steps; NOT the headline"}
    if seq is None:  # synthetic trajectory: known ground truth
        gt = np.linalg.inv(synth.trajectory_pose(k0)) @ synth.trajectory_pose(k0 + n_before + K - 1)  # pose taken before the host-vector stretch
        drift = float(np.linalg.norm((np.linalg.inv(pose) @ gt)[:3, 3]))
    else:
        drift = float("nan")

    # ---- accuracy of the whole run with the KITTI odometry devkit's metric as the reference vendors it
    #      (src/util/kitti_utils.cpp:108-191): relative translation / rotation error over all sub-sequences
    #      of 100 .. 800 m, from the pose table of the map (one pose per integrated scan) against the ground truth --
    #      the synthetic trajectory, or <dataset>/poses/XX.txt for a KITTI directory that has one
    odometry = None
    if rank == 0:
        try:
            from semantic_suma_amd import kitti as kitti_io
            est = pipe.map.poses().astype(np.float64)
            gt = None
            if seq is None:
                T0i = np.linalg.inv(synth.trajectory_pose(k0))
                gt = np.stack([T0i @ synth.trajectory_pose(k0 + k) for k in range(len(est))])
            else:
                pf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(kitti_dir))), "poses",
                                  os.path.basename(os.path.normpath(kitti_dir)) + ".txt")
                if os.path.exists(pf) and len(est) <= len(seq):
                    g = kitti_io.read_poses(pf, seq.calib.get("Tr"))
                    gt = np.stack([np.linalg.inv(g[k0]) @ g[k0 + k] for k in range(len(est))])
            e = kitti_io.odometry_errors(gt, est) if gt is not None else None
            if e is not None:
                odometry = {"t_err_percent": round(100.0 * e["t_err"], 4), "r_err_deg_per_100m": round(np.degrees(e["r_err"]) * 100.0, 4),
                            "segments": e["segments"], "scans": int(len(est)),
                            "trajectory_m": round(float(np.linalg.norm(np.diff(gt[:, :3, 3], axis=0), axis=1).sum()), 1),
                            "metric": "KITTI odometry devkit (kitti_utils.cpp:108-191): mean over all sub-sequences of 100..800 m, every 10th frame"}
        except Exception as e:  # noqa: BLE001 -- an extra must never cost the bench line
            print(f"odometry error metric not taken: {e!r}", file=sys.stderr)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "scans_per_sec", "value": world * K / elapsed, "unit": "scans/s", "n_gpus": world, "steps": K,
        "warmup": Wu, "ms_per_step": 1000.0 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "kitti" if seq is not None else "synthetic", **args.dist_info,
        "config": {"workload": f"BASELINE configs[1]: synthetic KITTI-like sequence, {H}x{W} range images, "
                               f"semantic-weighted ICP ({args.icp_iterations} GN iterations + stats pass) + surfel "
                               "fusion, one sequence per GPU, scans resident in HBM, processScan called scan after scan by the "
                               "native host loop (suma_pipeline_run_scans); timed after "
                               f"{n_before} untimed scans of the same sequence ({Wu} warm-up + pre-roll to the steady "
                               f"map size: {map_size_start} surfels at the start of the timed region)",
                   "points_per_scan": round(n_points), "preroll_scans": n_before, "map_surfels_start": map_size_start,
                   "map_surfels_end": map_size, "drift_m": round(drift, 4), "odometry_error": odometry,
                   "parallelism": f"sequence-sharded x{world}" if world > 1 else "single GPU"},
    }
    # host time of the K calls of the timed region: a stall of the calling thread (seen about once in five 60-scan runs on
    # the pool's boxes, ~1 ms) is one long call, not a slower kernel
    tc = timed_call_s * 1e6
    out["timed_call_us"] = {"median": round(float(np.median(tc)), 1), "max": round(float(tc.max()), 1), "argmax": int(tc.argmax()),
                            "calls_over_twice_the_median": int((tc > 2.0 * np.median(tc)).sum())}
    if per_rank is not None:
        out["per_rank"] = per_rank
    if cold_start is not None:
        out["cold_start"] = cold_start  # the first K scans of the sequence (growing map): NOT the headline
    if host_vectors is not None:
        out["host_vector_entry"] = host_vectors
    if loop_closure is not None:
        out["loop_closure_verify"] = loop_closure
    if reference_mode is not None:
        out["reference_mode"] = reference_mode

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

    # ---- CPU baseline: the oracle (port of the reference path) on THE SAME SCANS as the timed region -- it replays the
    #      sequence from its first scan (untimed, OpenMP) so that its map is the steady-state map the GPU number is
    #      taken on (round 3 timed the oracle on the first, cold scans), then the sample is timed.  Built -O3
    #      -march=native on this very host (bit-identical to the -O2 build the tests use:
    #      tests/test_oracle_kat.py::test_native_build_is_bit_identical)
    if world == 1 and args.cpu_scans > 0:
        from oracle import pyoracle
        variant = "native" if pyoracle.build_native() else ""
        ncores = os.cpu_count() or 1
        # tools/oracle_scaling.py on the 256-core MI355X host: 8 -> 29, 16 -> 48, 32 -> 36, 64 -> 24 scans/s.  Round 5 found
        # out why it peaks at 16: the box's cgroup grants 16 CPUs of the 256 it shows (cpu.max = 1600000 100000); more
        # threads only get throttled.
        threads = max(1, min(ncores, 16))
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else round(float(q) / float(per), 2)
        except (OSError, ValueError):
            pass
        n_cpu = min(args.cpu_scans, K)
        n_cpu1 = max(1, min(n_cpu // 4, HV if HV else n_cpu // 4))  # one thread: a shorter stretch right behind it
        op = pyoracle.OraclePipeline(p, variant=variant, threads=threads)
        tpre = time.perf_counter()
        for k in range(n_before):
            op.process_scan(*scans[k][4], fixed_iterations=args.icp_iterations)
        tpre = time.perf_counter() - tpre
        tc = time.perf_counter()
        for k in range(n_before, n_before + n_cpu):
            op.process_scan(*scans[k][4], fixed_iterations=args.icp_iterations)
        tc = time.perf_counter() - tc
        same_pose = bool(np.array_equal(op.pose(), pose)) if n_cpu == K else None
        flags = "-O3 -march=native" if variant else "-O2"
        out["cpu_baseline"] = {"value": n_cpu / tc, "unit": "scans/s", "cores": threads, "kind": "port",
                               "sample": f"scans {n_before}..{n_before + n_cpu - 1} of the same {H}x{W} sequence -- the first "
                                         f"{n_cpu} scans of the timed region, on the same steady-state map (the oracle replayed "
                                         f"scans 0..{n_before - 1} first, untimed, {tpre:.1f} s) -- through oracle/ (gcc {flags}), "
                                         f"OpenMP, {threads} threads of {ncores} host cores",
                               "map_surfels": int(op.ctx.map_size()), "pose_bits_equal_gpu": same_pose,
                               "host_cores_visible": ncores, "cgroup_cpu_quota_cores": quota}
        if n_before + n_cpu + n_cpu1 <= len(scans) and scans[n_before + n_cpu + n_cpu1 - 1][4] is not None:
            op.ctx.set_threads(1)
            tc = time.perf_counter()
            for k in range(n_before + n_cpu, n_before + n_cpu + n_cpu1):
                op.process_scan(*scans[k][4], fixed_iterations=args.icp_iterations)
            tc = time.perf_counter() - tc
            out["cpu_baseline_single_thread"] = {"value": n_cpu1 / tc, "unit": "scans/s", "cores": 1, "kind": "port",
                                                 "sample": f"the next {n_cpu1} scans ({n_before + n_cpu}..) of the same run, one thread"}
        del op
        # ---- optional: the LITERAL baseline of north_star, "the reference's own path timed on the host CPU" -- the
        #      reference's GLSL (read where it lies: SUMA_REFERENCE_SHADERS or /root/reference/src/shader) executed by
        #      Mesa llvmpipe, host code in numpy (tools/reference_gl_baseline.py).  Only where both exist (not on the
        #      driver's GPU box); a few scans on the same steady map.  Context, never credit.
        if args.reference_gl_scans > 0 and os.path.isdir(os.environ.get("SUMA_REFERENCE_SHADERS", "/root/reference/src/shader")):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("reference_gl_baseline", os.path.join(ROOT, "tools", "reference_gl_baseline.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                rg = mod.measure(preroll=n_before, scans=args.reference_gl_scans, get_scan=lambda k: scans[k][4])
                if rg is not None:
                    out["cpu_baseline_reference_gl"] = rg
            except Exception as e:  # noqa: BLE001 -- an optional leg must never cost the bench line
                print(f"cpu_baseline_reference_gl not taken: {e!r}", file=sys.stderr)
    # ---- what a host pays that does NOT use the scan pipeline: the class-by-class call sequence of the reference's
    #      processScan on the adapter classes (include/suma_adapter.hpp), the phase API with (empty) loop-closure hooks,
    #      and the one-call pipeline -- one C++ driver, the same host vectors (tools/adapter_bench.cpp)
    if world == 1 and args.adapter_scans > 0:
        # the same stretch of the sequence as the timed region: the driver replays the pre-roll (untimed) in every mode
        n_ad = min(args.adapter_scans, K + HVW + HV)
        ap_res = adapter_path(scans[:n_before + n_ad], W, H, args.icp_iterations, preroll=n_before)
        if ap_res is not None:
            out["adapter_path"] = ap_res
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
