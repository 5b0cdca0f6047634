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


def hypothesis_starts(T0: np.ndarray, n: int, seed: int = 1234, max_t: float = 0.2, max_deg: float = 2.0):
    """n start poses T0 * exp(xi_k), xi_k ~ U(+-max_t m, +-max_deg deg) with seed 1234 + k; k = 0 unperturbed
    (SURVEY.md 8d config 3).  Deterministic and identical on every rank."""
    out = []
    for k in range(n):
        T = np.array(T0, dtype=np.float64)
        if k:
            rng = np.random.default_rng(seed + k)
            t = rng.uniform(-max_t, max_t, 3)
            w = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
            th = float(np.linalg.norm(w))
            K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            R = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
            D = np.eye(4)
            D[:3, :3], D[:3, 3] = R, t
            T = T @ D
        out.append(T)
    return out


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
