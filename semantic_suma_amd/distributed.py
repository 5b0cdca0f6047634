"""Multi-GPU sharding of the hot path: replicas only.

Within one sequence every scan depends on the pose and the map of the previous scan
(reference src/core/SurfelMapping.cpp:453-457, 799), so the path shards across INDEPENDENT units:
  * sequences  (BASELINE config 4): one full pipeline per GPU, no data-path collective;
  * hypotheses (BASELINE config 3, the reference's loop-closure verification pattern
    SurfelMapping.cpp:662-779): every rank minimises the same frame pair from its own start pose.
The only exchange is a gather of poses (+ 5 statistics) -- one small all_gather over RCCL/xGMI
(backend "nccl" is RCCL on ROCm) or gloo on CPU for the tests.  One process per GPU.
"""
from __future__ import annotations

import numpy as np


def lpt_assign(lengths, world_size: int):
    """Longest-processing-time-first assignment of sequences to ranks.
    Returns (assignment[rank] -> list of sequence ids, load[rank])."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        out[r].append(i)
        loads[r] += lengths[i]
    return out, loads


def hypothesis_perturbations(n: int, seed: int = 1234, max_t: float = 0.2, max_deg: float = 2.0):
    """n rigid motions D_k = exp(xi_k), xi_k ~ U(+-max_t m, +-max_deg deg) with seed 1234 + k; D_0 = identity
    (SURVEY.md 8d config 3).  Deterministic and identical on every rank."""
    out = []
    for k in range(n):
        D = np.eye(4)
        if k:
            rng = np.random.default_rng(seed + k)
            t = rng.uniform(-max_t, max_t, 3)
            w = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
            th = float(np.linalg.norm(w))
            K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            R = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
            D[:3, :3], D[:3, 3] = R, t
        out.append(D)
    return out


def hypothesis_starts(T0: np.ndarray, n: int, seed: int = 1234, max_t: float = 0.2, max_deg: float = 2.0):
    """n start poses T0 * D_k (hypothesis_perturbations), formed in the fixed operation order of mul4 -- the native
    runner (suma_run_hypotheses) forms the very same products, k = 0 included."""
    T0 = np.array(T0, dtype=np.float64)
    return [mul4(T0, D) for D in hypothesis_perturbations(n, seed, max_t, max_deg)]


def pick_winner(stats) -> int:
    """smallest residual per valid pair; ties -> lowest hypothesis index (same decision on every rank)"""
    best, best_v = 0, None
    for k, s in enumerate(stats):
        v = float("inf") if s[1] <= 0 else s[0] / s[1]
        if best_v is None or v < best_v:
            best, best_v = k, v
    return best


def gather_poses(local: np.ndarray, device=None) -> np.ndarray:
    """all_gather of a float64 array of identical shape on every rank -> [world, ...].
    The one collective of the data path; called once per scan (hypotheses) or once per job (sequences)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(local, dtype=np.float64)[None]
    t = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return np.stack([o.cpu().numpy() for o in outs])


class NativeGather:
    """The gather of SURVEY.md 8(b) -- `suma_gather_poses` / `suma_gather` of libsuma_hip_dist.so: one RCCL all-gather on
    the ctx stream -- for a job whose ranks were started by torch.distributed.  torch.distributed only BOOTSTRAPS it (rank
    0's 128-byte RCCL id is broadcast over the existing process group, as a C++ host would send it over MPI or a socket)
    and carries the launcher's barrier; the data-path collective is the library's own.  Every rank must construct it at
    the same point.  `ok` is False (with `error`) on every rank if any rank failed to create its communicator."""

    def __init__(self, ctx, device=None):
        import ctypes as C
        import os

        import torch
        import torch.distributed as dist

        from . import core
        self.ctx, self.comm, self.ok, self.error = ctx, None, False, ""
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        vp = C.c_void_p
        uid = torch.zeros(128, dtype=torch.uint8, device=device)
        flag = 1
        try:
            D = C.CDLL(os.path.join(os.path.dirname(core.LIB_PATH), "libsuma_hip_dist.so"))
            D.suma_dist_unique_id.argtypes = [vp]
            D.suma_dist_comm_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
            D.suma_dist_comm_destroy.argtypes = [vp]
            D.suma_gather_poses.argtypes = [vp, vp, vp, vp]
            D.suma_gather.argtypes = [vp, vp, vp, C.c_uint32, vp]
            D.suma_dist_last_error.restype = C.c_char_p
            D.suma_dist_last_error.argtypes = [vp]
            self.D = D
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                if D.suma_dist_unique_id(buf) != 0:
                    raise RuntimeError("suma_dist_unique_id: " + D.suma_dist_last_error(None).decode())
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(uid.device)
        except (OSError, RuntimeError) as e:  # the library is missing or RCCL refused: every rank must learn it
            flag, self.error = 0, repr(e)
        ok = torch.tensor([flag], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # nobody enters ncclCommInitRank unless everybody can
        if int(ok.item()) == 0:
            self.error = self.error or "another rank could not load libsuma_hip_dist.so"
            return
        dist.broadcast(uid, src=0)
        comm = vp()
        buf = C.create_string_buffer(bytes(uid.cpu().numpy().tobytes()), 128)
        rc = self.D.suma_dist_comm_create(buf, self.world, self.rank, C.byref(comm))
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if rc != 0:
            self.error = "suma_dist_comm_create: " + self.D.suma_dist_last_error(None).decode()
        elif int(ok.item()) == 0:
            self.error = "suma_dist_comm_create failed on another rank"
            self.D.suma_dist_comm_destroy(comm)
        else:
            self.comm, self.ok = comm, True

    def gather(self, local: np.ndarray) -> np.ndarray:
        """all-gather of <= 64 doubles per rank behind everything enqueued on the ctx -> [world, ...]"""
        a = np.ascontiguousarray(local, dtype=np.float64)
        out = np.zeros((self.world,) + a.shape, dtype=np.float64)
        rc = (self.D.suma_gather_poses(self.ctx.h, self.comm, a.ctypes.data, out.ctypes.data) if a.size == 16 else
              self.D.suma_gather(self.ctx.h, self.comm, a.ctypes.data, a.size, out.ctypes.data))
        if rc != 0:
            raise RuntimeError("suma_gather: " + self.D.suma_dist_last_error(self.comm).decode())
        return out

    def close(self):
        if self.comm is not None:
            self.D.suma_dist_comm_destroy(self.comm)
            self.comm = None


# ---------------------------------------------------------------------------------------------------------------
# GPU-side runners for BASELINE configs 3 and 4.  They are written against a small engine interface so that the very
# same orchestration code runs over the HIP path (HipEngine below) and, in the world-2 gloo tests on CPU, over a
# stand-in engine built by the test.
# ---------------------------------------------------------------------------------------------------------------

def conf_threshold(params, t: int) -> float:
    """SurfelMapping::getConfidenceThreshold (SurfelMapping.cpp:333-340, time_init = 10)"""
    f = np.float32
    ct = f(params.confidence_threshold)
    if t < 10:
        pu = f(0.1)
        log_unstable = f(np.log(float(pu / (f(1.0) - pu))))
        alpha = f(t) / f(10)
        ct = f((1.0 - float(alpha)) * float(log_unstable) + float(alpha * f(params.confidence_threshold)))
    return float(ct)


def mul4(A, B):
    """4x4 product in a fixed operation order (numpy's matmul may reorder / fuse): every rank must reach the same bits"""
    C = np.zeros((4, 4))
    for r in range(4):
        for c in range(4):
            C[r, c] = ((A[r, 0] * B[0, c] + A[r, 1] * B[1, c]) + A[r, 2] * B[2, c]) + A[r, 3] * B[3, c]
    return C


class HipEngine:
    """One GPU's share of the hypothesis runner on the HIP path: Preprocessing, SurfelMap, Frame2Model and the
    batched device-resident Gauss-Newton (suma_icp_minimize_batch) of one context."""

    def __init__(self, params, device: int = 0):
        from . import core
        self.core = core
        self.params = params
        self.ctx = core.Context(params, device)
        self.pre = core.Preprocessing(self.ctx)
        self.map = core.SurfelMap(self.ctx)
        self.objective = core.Frame2Model(self.ctx)
        self.gn = core.LieGaussNewton(self.ctx)
        self.current = core.Frame(self.ctx, params.data_width, params.data_height)
        self.model = core.Frame(self.ctx, params.model_width, params.model_height)

    def preprocess(self, points, labels, probs, t):
        self.pre.process(points, self.current, labels, probs, t)

    def render(self, pose, ct):
        self.map.render(pose, pose, self.model, ct)

    def minimize(self, starts):
        """-> (poses [n, 4, 4], [(error, valid, outlier)] per start) against the rendered model"""
        self.objective.setData(self.current, self.map.newMapFrame())
        poses, stats = self.gn.minimize_batch(starts, self.objective)
        return poses, [(s["error"], s["valid"], s["outlier"]) for s in stats]

    def update(self, pose):
        self.map.update(pose, self.current)

    def map_bytes(self) -> bytes:
        return self.map.getAllSurfels().tobytes()


def run_hypotheses(engine, scans, n_hyp: int, rank: int = 0, world: int = 1, gather=gather_poses, on_scan=None):
    """BASELINE config 3: per scan, n_hyp Gauss-Newton chains from perturbed starts (hypothesis k on rank k % world),
    ONE all-gather of the results (18 doubles per hypothesis), the same winner on every rank, and the map update with
    the winner's pose on every rank (maps stay identical without any map traffic).
    Preprocessing and model rendering are done redundantly per rank (no traffic).
    Returns (poses per scan [n, 4, 4], winners per scan)."""
    params = engine.params
    pose, increment = np.eye(4), np.eye(4)
    poses, winners = [], []
    for t, (pts, lab, prob) in enumerate(scans):
        engine.preprocess(pts, lab, prob, t)
        ct = conf_threshold(params, t)
        engine.render(pose, ct)
        if t > 0:
            starts = hypothesis_starts(increment, n_hyp)
            mine = list(range(rank, n_hyp, world))
            local = np.zeros((n_hyp, 18))
            if mine:
                Ts, stats = engine.minimize([starts[k] for k in mine])
                for j, k in enumerate(mine):
                    local[k, :16] = np.asarray(Ts[j]).ravel()
                    local[k, 16], local[k, 17] = stats[j][0], stats[j][1]
            allr = gather(local).sum(axis=0) if world > 1 else local  # every hypothesis is owned by exactly one rank
            win = pick_winner([(allr[k, 16], allr[k, 17]) for k in range(n_hyp)])
            increment = allr[win, :16].reshape(4, 4).copy()
            pose = mul4(pose, increment)
            winners.append(win)
        else:
            winners.append(-1)
        engine.update(pose)
        poses.append(pose.copy())
        if on_scan is not None:
            on_scan(t, pose)
    return np.stack(poses), winners


def run_sequences(my_sequences, make_pipeline, scans_of, fixed_iterations: int = 0, threads: bool = True):
    """BASELINE config 4: the sequences assigned to this rank (lpt_assign), each through its own pipeline
    (make_pipeline() -> object with processScan(points, labels, probs, fixed_iterations) and getCurrentPose()).
    A rank that owns several sequences runs them as concurrent pipelines on separate host threads and HIP streams
    of its GPU (one context each): the single-sequence pipeline is latency bound and leaves most CUs idle.
    Returns {sequence id: (n_scans, final pose)}; no collective in here."""
    import threading
    out, errors = {}, []

    def run(seq):
        try:
            pipe = make_pipeline()
            n = 0
            for pts, lab, prob in scans_of(seq):
                pipe.processScan(pts, lab, prob, fixed_iterations=fixed_iterations)
                n += 1
            out[seq] = (n, pipe.getCurrentPose())
        except Exception as e:  # noqa: BLE001 -- reported to the caller below
            errors.append((seq, repr(e)))

    if threads and len(my_sequences) > 1:
        th = [threading.Thread(target=run, args=(s,)) for s in my_sequences]
        [t.start() for t in th]
        [t.join() for t in th]
    else:
        for s in my_sequences:
            run(s)
    if errors:
        raise RuntimeError(f"sequence runs failed: {errors}")
    return out


# ---------------------------------------------------------------------------------------------------------------
# The same two runners with the host loop in C++ (include/suma_runner.h, csrc/suma_runner.hip): no interpreter between
# two scans.  The Python loops above stay as the reference orchestration (the world-2 gloo tests drive them over a
# stand-in engine on CPU); tests/test_gpu_dist.py demands identical poses, winners and maps from both.
# ---------------------------------------------------------------------------------------------------------------

def run_hypotheses_hip(params, scans, n_hyp: int, rank: int = 0, world: int = 1, device: int = 0,
                       fixed_iterations: int = 0, gather=gather_poses, on_device: bool = False):
    """BASELINE config 3 through suma_run_hypotheses.  One exchange per scan: every rank's [n_hyp, 18] table is
    gathered with `gather` and summed (each row is owned by exactly one rank).  Returns (poses [n, 4, 4], winners)."""
    from . import core
    exchange = (lambda local: gather(local).sum(axis=0)) if world > 1 else None
    return core.run_hypotheses_native(params, scans, hypothesis_perturbations(n_hyp), rank, world, device,
                                      fixed_iterations, exchange, on_device)


def run_sequences_hip(params, sequences, device: int = 0, fixed_iterations: int = 0, max_concurrent: int = 4,
                      on_device: bool = False):
    """BASELINE config 4, one rank's share, through suma_run_sequences: `sequences` = {sequence id: list of scans};
    the longest sequence starts first (LPT inside the rank), at most max_concurrent pipelines at a time.
    Returns {sequence id: (n_scans, final pose)} like run_sequences."""
    from . import core
    order = sorted(sequences, key=lambda s: (-len(sequences[s]), s))
    res = core.run_sequences_native(params, [sequences[s] for s in order], device, fixed_iterations, max_concurrent,
                                    on_device)
    return {s: (r["scans_done"], r["end_pose"]) for s, r in zip(order, res)}
