"""Multi-GPU path on CPU: world_size-2 `gloo` runs of the sharding + pose-gather logic
(semantic_suma_amd/distributed.py).  There is no GPU here, so the per-rank compute is the CPU oracle
(test infrastructure standing in for the HIP pipeline, which has the same interface and -- by the GPU
parity tests -- the same bits); what is under test is the distribution: LPT sequence assignment,
hypothesis sharding, the single all_gather of poses and the rank-consistent winner selection."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 180, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import pyoracle
    from semantic_suma_amd import synth
    from semantic_suma_amd.distributed import gather_poses, hypothesis_starts, lpt_assign, pick_winner
    from semantic_suma_amd.types import params_with_size
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = params_with_size(W, H, max_iterations=6)
    try:
        if mode == "sequences":
            lengths = [3, 1, 2]  # scans per sequence
            assign, loads = lpt_assign(lengths, world)
            poses = np.zeros((len(lengths), 4, 4))
            for seq in assign[rank]:
                pipe = pyoracle.OraclePipeline(p)
                for k in range(lengths[seq]):
                    pts, lab, prob, _ = synth.generate_scan(100 * seq + k, n_azimuth=W, height=H)
                    pipe.process_scan(pts, lab, prob, fixed_iterations=6)
                poses[seq] = pipe.pose()
            allp = gather_poses(poses)  # [world, n_seq, 4, 4]; one collective for the whole job
            merged = allp.sum(axis=0)   # every sequence is owned by exactly one rank
            q.put((rank, assign, loads, merged))
        else:
            ora = pyoracle.Oracle(p)
            s0 = synth.generate_scan(0, n_azimuth=W, height=H)
            s1 = synth.generate_scan(1, n_azimuth=W, height=H)
            f0 = ora.preprocess(*s0[:3], 20, ora.frame())
            f1 = ora.preprocess(*s1[:3], 21, ora.frame())
            T0 = np.eye(4)
            T0[0, 3] = 1.0
            starts = hypothesis_starts(T0, 4)
            local = np.zeros((4, 18))
            for k in range(rank, 4, world):  # hypothesis k lives on rank k % world
                T, _, st = ora.minimize(f1, f0, starts[k])
                local[k, :16], local[k, 16], local[k, 17] = T.ravel(), st.error, st.valid
            allr = gather_poses(local).sum(axis=0)
            win = pick_winner([(allr[k, 16], allr[k, 17]) for k in range(4)])
            q.put((rank, win, allr))
    finally:
        dist.destroy_process_group()


def _run(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    out = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return sorted(out, key=lambda o: o[0])


def test_lpt_assignment():
    from semantic_suma_amd.distributed import lpt_assign
    kitti = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]  # odometry 00-10 scan counts
    assign, loads = lpt_assign(kitti, 8)
    assert sorted(i for a in assign for i in a) == list(range(11))
    assert max(loads) == 4661  # makespan bound by the longest sequence
    assert lpt_assign([5], 4)[0] == [[0], [], [], []]


def test_sequence_sharding_gloo_world2(oracle_lib):
    (r0, a0, l0, m0), (r1, a1, l1, m1) = _run("sequences")
    assert a0 == a1 and l0 == l1 and sorted(a0[0] + a0[1]) == [0, 1, 2]
    assert np.array_equal(m0, m1)  # both ranks end with the same gathered trajectory endpoints
    assert all(abs(np.linalg.det(m0[s][:3, :3]) - 1) < 1e-9 for s in range(3))
    assert abs(m0[0][0, 3]) > 0.5  # sequence 0 (3 scans) moved


def test_hypothesis_sharding_gloo_world2(oracle_lib):
    (r0, w0, a0), (r1, w1, a1) = _run("hypotheses")
    assert w0 == w1 and np.array_equal(a0, a1)
    assert (a0[:, 17] > 0).all()  # every hypothesis was evaluated by exactly one rank
    best = min(range(4), key=lambda k: a0[k, 16] / a0[k, 17])
    assert w0 == best
