"""GPU tests of the multi-GPU runners on ONE device (run with -m gpu): two processes (gloo rendezvous, both on
device 0) drive the HIP path through semantic_suma_amd.distributed.run_hypotheses and must reach, rank for rank, the
winners / poses / map of a single process that runs all hypotheses itself; config 4's runner with three
sequences as concurrent pipelines on one GPU; a world-1 RCCL process group with its collectives running beside the
pipeline in one process (what each rank of bench.py --gpus N does).  RCCL between ranks needs one GPU per rank: bench.py --gpus N."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

from semantic_suma_amd.types import params_with_size

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, N_HYP, N_SCANS = 900, 8, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from semantic_suma_amd import synth
    from semantic_suma_amd.distributed import HipEngine, run_hypotheses
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = params_with_size(W, max_iterations=8)
        scans = [synth.generate_scan(k, n_azimuth=W)[:3] for k in range(N_SCANS)]
        eng = HipEngine(p, device=0)
        poses, winners = run_hypotheses(eng, scans, N_HYP, rank, world)
        q.put((rank, world, poses, winners, hashlib.sha256(eng.map_bytes()).hexdigest()))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_hypotheses_two_ranks_one_device_equal_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    procs.append(ctx.Process(target=_worker, args=(0, 1, _free_port(), q)))
    for pr in procs:
        pr.start()
    out = [q.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    ref = [o for o in out if o[1] == 1][0]
    for o in out:
        assert np.array_equal(o[2], ref[2]), f"rank {o[0]} of {o[1]}: poses differ from the single-process run"
        assert o[3] == ref[3] and o[4] == ref[4]
    assert ref[3][0] == -1 and all(0 <= w < N_HYP for w in ref[3][1:])
    assert 3.0 < ref[2][-1][0, 3] < 6.0  # ~1.1 m per scan


def test_sequences_as_concurrent_pipelines():
    from semantic_suma_amd import core, synth
    from semantic_suma_amd.distributed import lpt_assign, run_sequences
    p = params_with_size(W)
    lengths = [6, 4, 3]
    assign, loads = lpt_assign(lengths, 1)

    def scans_of(seq):
        for k in range(lengths[seq]):
            yield synth.generate_scan(50 * seq + k, n_azimuth=W)[:3]

    a = run_sequences(assign[0], lambda: core.SurfelMapping(p), scans_of, fixed_iterations=10, threads=True)
    b = run_sequences(assign[0], lambda: core.SurfelMapping(p), scans_of, fixed_iterations=10, threads=False)
    assert sorted(a) == [0, 1, 2]
    for s in a:
        assert a[s][0] == lengths[s] and np.array_equal(a[s][1], b[s][1]), f"sequence {s}"


def test_native_sequence_runner_equals_the_python_runner():
    """suma_run_sequences (C++ host loop, one thread per pipeline) against run_sequences, from host arrays (staged two
    scans ahead) and from scans resident in HBM"""
    from semantic_suma_amd import core, synth
    from semantic_suma_amd.distributed import run_sequences, run_sequences_hip
    p = params_with_size(W)
    lengths = {0: 6, 1: 4, 2: 3, 3: 5}
    seqs = {s: [synth.generate_scan(50 * s + k, n_azimuth=W)[:3] for k in range(n)] for s, n in lengths.items()}
    ref = run_sequences(list(lengths), lambda: core.SurfelMapping(p), lambda s: seqs[s], fixed_iterations=10, threads=False)
    a = run_sequences_hip(p, seqs, fixed_iterations=10, max_concurrent=3)
    ctx = core.Context(p)
    dev = {s: [tuple(ctx.device_array(x) for x in sc) + (sc[0].shape[0],) for sc in seqs[s]] for s in seqs}
    b = run_sequences_hip(p, dev, fixed_iterations=10, max_concurrent=2, on_device=True)
    for s, n in lengths.items():
        assert a[s][0] == n and b[s][0] == n
        assert np.array_equal(a[s][1], ref[s][1]), f"sequence {s}: host arrays"
        assert np.array_equal(b[s][1], ref[s][1]), f"sequence {s}: resident scans"


def test_native_hypothesis_runner_equals_the_python_runner():
    """suma_run_hypotheses (C++ host loop on the scan pipeline's phases) against run_hypotheses over HipEngine: the same
    poses and winners from one rank, and from two ranks that exchange their tables (two threads, one device)"""
    import threading
    from semantic_suma_amd import synth
    from semantic_suma_amd.distributed import HipEngine, run_hypotheses, run_hypotheses_hip
    p = params_with_size(W, max_iterations=8)
    scans = [synth.generate_scan(k, n_azimuth=W)[:3] for k in range(N_SCANS)]
    ref_poses, ref_winners = run_hypotheses(HipEngine(p, device=0), scans, N_HYP)
    poses, winners = run_hypotheses_hip(p, scans, N_HYP)
    assert winners == ref_winners and np.array_equal(poses, ref_poses)
    # two ranks in one process: the exchange is a barrier + sum of the two tables
    tables, out, barrier = {}, {}, threading.Barrier(2)

    def exchange_for(rank):
        def gather(local):
            tables[rank] = np.array(local)
            barrier.wait()
            tot = np.stack([tables[0], tables[1]])
            barrier.wait()
            return tot
        return gather

    def run(rank):
        out[rank] = run_hypotheses_hip(p, scans, N_HYP, rank=rank, world=2, gather=exchange_for(rank))

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    for r in range(2):
        assert out[r][1] == ref_winners and np.array_equal(out[r][0], ref_poses), f"rank {r}"


def test_bench_contract_with_two_ranks_on_one_device():
    """bench.py's N > 1 path end to end -- torch.distributed.run, one process per rank, barrier + max-over-ranks timing,
    the pose gather, rank 0's JSON line -- with two ranks sharing this box's one GPU (gloo; RCCL needs a GPU per rank)"""
    import json
    import subprocess
    env = dict(os.environ, SUMA_BENCH_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "2", "--backend", "gloo", "--cpu-scans", "0", "--no-kernel-events", "--preroll", "0", "--adapter-scans", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 100 and abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6 / 1000.0)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "k6_icp_step" and "cpu_baseline" not in d
    assert d["config"]["drift_m"] < 0.5


def test_bench_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with NO external launcher: bench.py starts torch.distributed.run itself, two ranks
    meet in a process group (gloo here: this box has one GPU, RCCL wants one per rank) and rank 0 reports n_gpus = 2"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SUMA_BENCH_FORCE_DEVICE"] = "0"
    for mode, extra in (("single", ["--preroll", "0"]), ("hypotheses", []), ("sequences11", [])):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--backend",
               "gloo", "--cpu-scans", "0", "--no-kernel-events", "--adapter-scans", "0", "--mode", mode,
               "--width", "900"] + extra
        out = subprocess.run(cmd, env=dict(env, SUMA_SEQ_SCALE="400"), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, mode + out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["devices"] == [0], (mode, d)
        assert d["value"] > 10


def test_rccl_process_group_beside_the_pipeline():
    """What every rank of `bench.py --gpus N` (N > 1) does, with N = 1: a torch.distributed process group on RCCL
    (backend "nccl") lives in the same process as libsuma_hip.so, collectives on CUDA tensors (barrier, all_reduce
    MAX, all_gather of poses) run between scans, the native pose gather (libsuma_hip_dist.so, bootstrapped over the
    process group) returns the pose, and the pipeline's results do not change.  Fresh interpreter, torch
    first -- the order bench.py uses (one HIP runtime in the process)."""
    import subprocess
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{_free_port()}", RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from semantic_suma_amd import core, synth
from semantic_suma_amd.types import params_with_size
p = params_with_size({W})
scans = [synth.generate_scan(k, n_azimuth={W})[:3] for k in range(4)]
solo = core.SurfelMapping(p, device=0)
for s in scans: solo.processScan(*s, fixed_iterations=8)
want = solo.getCurrentPose().copy()
dist.init_process_group("nccl", device_id=dev)
pipe = core.SurfelMapping(p, device=0)
for s in scans:
    pipe.processScan(*s, fixed_iterations=8)
    dist.barrier()
    t = torch.tensor([float(pipe.map.size())], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert int(t.item()) == pipe.map.size()
    x = torch.as_tensor(pipe.getCurrentPose().copy()).to(dev)
    outs = [torch.empty_like(x)]
    dist.all_gather(outs, x)
    assert np.array_equal(outs[0].cpu().numpy(), pipe.getCurrentPose())
# the pose gather bench.py --gpus N uses for N > 1: suma_gather_poses of libsuma_hip_dist.so (the SURVEY 8(b) export),
# bootstrapped over this process group -- here with the one rank a 1-GPU box can host
from semantic_suma_amd.distributed import NativeGather
ng = NativeGather(pipe.ctx, device=dev)
assert ng.ok, ng.error
g = ng.gather(pipe.getCurrentPose())
assert g.shape == (1, 4, 4) and np.array_equal(g[0], pipe.getCurrentPose())
assert np.array_equal(ng.gather(np.arange(5.0))[0], np.arange(5.0))
ng.close()
torch.cuda.synchronize()
assert np.array_equal(pipe.getCurrentPose(), want), "pose changed beside the process group"
assert pipe.map.getAllSurfels().tobytes() == solo.map.getAllSurfels().tobytes()
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available())
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_OK True" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
