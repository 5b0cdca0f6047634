"""Deterministic synthetic LiDAR scans for tests and benchmarks (no KITTI data ships with this repo).

World and beam model restate the recipe of the reference's ``SimulationReader``
(reference src/io/SimulationReader.cpp:68-177: ground plane, 23 cubes, HDL-64E beam table
``90-2+i/3`` / ``90+8.83+i/2`` deg) extended as SURVEY.md 8(d) prescribes: two long walls at
y = +-25 m for yaw / x observability, semantic labels (ground = 40 road, cubes = 10 car,
walls = 50 building), half of the cubes moving 0.5 m per scan, range noise sigma = 0.02 m.

This is an input generator only: it produces ``rv::Laserscan``-shaped data
(N x 4 float32 points ``x, y, z, 1`` as ``rv::Point3f`` src/rv/geometry.h:331-345, ``labels_float``,
``labels_prob`` as in src/io/KITTIReader.cpp:175-200).  Pure numpy, runs anywhere.
"""
from __future__ import annotations

import math

import numpy as np

SENSOR_HEIGHT = 1.73
WALL_Y = 25.0
LABEL_ROAD, LABEL_CAR, LABEL_BUILDING = 40.0, 10.0, 50.0

# (x, y, z, yaw_deg, size) of the 23 cubes of SimulationReader.cpp:72-99 (roll of the last cube dropped)
_CUBES = [
    (10, 10, 0.5, 0.0, 1.0), (112, -15, 1.25, 45.0, 2.5), (34, 20, 0.75, 0.0, 1.5), (50, -10, 0.75, 79.0, 1.5),
    (65, 5, 0.75, 45.0, 1.5), (70, -15, 0.85, 25.0, 1.5), (100, 30, 0.65, 0.0, 1.5), (120, -10, 0.65, 25.0, 1.5),
    (170, -10, 0.65, 15.0, 1.5), (190, -30, 0.65, 35.0, 1.5), (230, 15, 0.65, 5.0, 3.5), (270, -7, 0.65, -3.5, 2.5),
    (280, 20, 0.65, 2.0, 4.5), (320, 20, 0.65, 2.0, 4.5), (370, 10, 0.65, 15.0, 1.5), (390, -30, 0.65, 35.0, 1.5),
    (430, -15, 0.65, 5.0, 3.5), (470, 7, 0.65, -3.5, 2.5), (480, 20, 0.65, 2.0, 4.5), (40, -20, 0.65, 15.0, 3.0),
    (50, -75, 5.0, 25.0, 10.0), (-20, -54, 0.65, 15.0, 4.5), (150, 3, 1.0, 30.0, 2.0),
]


def beam_elevations(height: int) -> np.ndarray:
    """Elevation angles (rad, + up) of ``height`` beams: the HDL-64E table resampled to ``height`` rows."""
    up = 2.0 - np.arange(32) / 3.0                 # 90-2+i/3 deg from the z axis -> +2 .. -8.33 deg
    lo = -8.83 - np.arange(32) / 2.0               # -8.83 .. -24.33 deg
    table = np.concatenate([up, lo])               # descending
    if height == 64:
        el = table
    else:
        el = np.interp(np.linspace(0, 63, height), np.arange(64), table)
    return np.deg2rad(el).astype(np.float64)


def trajectory_pose(k: int, speed: float = 1.1) -> np.ndarray:
    """Sensor pose (4x4 float64, world <- sensor) of scan ``k`` on a stadium-shaped loop inside the walls."""
    L, R = 450.0, 12.0
    per = 2 * L + 2 * math.pi * R
    s = (k * speed) % per
    if s < L:                                      # +x straight at y = -R
        x, y, yaw = s, -R, 0.0
    elif s < L + math.pi * R:                      # right end, turning left
        a = (s - L) / R
        x, y, yaw = L + R * math.sin(a), -R * math.cos(a), a
    elif s < 2 * L + math.pi * R:                  # -x straight at y = +R
        x, y, yaw = L - (s - L - math.pi * R), R, math.pi
    else:
        a = (s - 2 * L - math.pi * R) / R
        x, y, yaw = -R * math.sin(a), R * math.cos(a), math.pi + a
    yaw += 0.01 * math.sin(0.37 * k)               # gentle wobble: keeps every GN dof excited
    c, sn = math.cos(yaw), math.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = [[c, -sn, 0], [sn, c, 0], [0, 0, 1]]
    T[:3, 3] = [x, y, SENSOR_HEIGHT + 0.02 * math.sin(0.11 * k)]
    return T


def _static_buildings():
    """Static 'building' blocks (label 50) along both walls: irregular spacing / depth, so that the
    translation along the corridor is observable from static structure (cars are dropped while
    timestamp < 10, gen_vertexmap.vert:95-102)."""
    rng = np.random.default_rng(2024)
    out = []
    x = -80.0
    while x < 540.0:
        for side in (-1.0, 1.0):
            depth = rng.uniform(2.0, 9.0)
            length = rng.uniform(3.0, 10.0)
            h = rng.uniform(3.0, 8.0)
            out.append((x + rng.uniform(-3, 3), side * (WALL_Y - 0.5 * depth), 0.5 * h, rng.uniform(-8, 8),
                        length, depth, h))
        x += rng.uniform(9.0, 17.0)
    return out


_BUILDINGS = _static_buildings()


def _boxes(k: int):
    """(centre[3], half[3], yaw, label) per box for scan k; odd cubes move 0.5 m per scan along x."""
    c = np.array([q[:3] for q in _CUBES], dtype=np.float64)
    h = np.array([[0.5 * q[4]] * 3 for q in _CUBES], dtype=np.float64)
    yaw = np.deg2rad(np.array([q[3] for q in _CUBES], dtype=np.float64))
    moving = (np.arange(len(_CUBES)) % 2) == 1
    c[moving, 0] = (c[moving, 0] + 0.5 * k + 60.0) % 560.0 - 60.0
    lab = np.full(len(_CUBES), LABEL_CAR)
    cb = np.array([q[:3] for q in _BUILDINGS], dtype=np.float64)
    hb = np.array([[0.5 * q[4], 0.5 * q[5], 0.5 * q[6]] for q in _BUILDINGS], dtype=np.float64)
    yb = np.deg2rad(np.array([q[3] for q in _BUILDINGS], dtype=np.float64))
    lb = np.full(len(_BUILDINGS), LABEL_BUILDING)
    return np.concatenate([c, cb]), np.concatenate([h, hb]), np.concatenate([yaw, yb]), np.concatenate([lab, lb])


def generate_scan(k: int, n_azimuth: int = 2000, height: int = 64, noise: float = 0.02, seed: int = 1337,
                  semantics: bool = True, max_range: float = 100.0, pose: np.ndarray | None = None):
    """Ray-cast scan ``k``.  Returns (points[N,4] f32, labels[N] f32, probs[N] f32, pose[4,4] f64)."""
    T = trajectory_pose(k) if pose is None else np.asarray(pose, dtype=np.float64)
    rng = np.random.default_rng(seed + 7919 * k)
    el = beam_elevations(height)
    az = (np.arange(n_azimuth) + rng.uniform(-0.3, 0.3, n_azimuth)) * (2 * math.pi / n_azimuth)
    # azimuth-major ordering, like SimulationReader's beam loop (SimulationReader.cpp:106-117)
    A, E = np.meshgrid(az, el, indexing="ij")
    A, E = A.ravel(), E.ravel()
    d_local = np.stack([np.cos(A) * np.cos(E), np.sin(A) * np.cos(E), np.sin(E)], axis=1)
    d = d_local @ T[:3, :3].T
    o = T[:3, 3]
    n_rays = d.shape[0]
    t_best = np.full(n_rays, np.inf)
    label = np.zeros(n_rays, dtype=np.float32)
    Tinv = np.linalg.inv(T)

    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(d[:, 2] < 0, -o[2] / d[:, 2], np.inf)           # ground z = 0
        hit = t < t_best
        t_best = np.where(hit, t, t_best)
        label = np.where(hit, LABEL_ROAD, label)
        for wy in (WALL_Y, -WALL_Y):                                 # walls, 6 m high
            t = (wy - o[1]) / d[:, 1]
            z = o[2] + t * d[:, 2]
            ok = (t > 0) & (z >= 0) & (z <= 6.0)
            t = np.where(ok, t, np.inf)
            hit = t < t_best
            t_best = np.where(hit, t, t_best)
            label = np.where(hit, LABEL_BUILDING, label)
        cz, half, yaw, blabel = _boxes(k)
        for b in range(cz.shape[0]):                                 # slab test in the box frame
            rel = Tinv[:3, :3] @ (cz[b] - o)                          # box centre in the sensor frame
            dist = math.hypot(rel[0], rel[1])
            rad = float(np.linalg.norm(half[b, :2]))
            if dist - rad > max_range:
                continue
            if dist <= rad * 1.05:
                cols = np.arange(n_azimuth)
            else:                                                    # only azimuth columns that can see the box
                bearing = math.atan2(rel[1], rel[0]) % (2 * math.pi)
                hw = math.asin(min(1.0, rad / dist)) + 2.0 * (2 * math.pi / n_azimuth)
                i0 = int(math.floor((bearing - hw) / (2 * math.pi) * n_azimuth))
                i1 = int(math.ceil((bearing + hw) / (2 * math.pi) * n_azimuth))
                cols = np.arange(i0, i1 + 1) % n_azimuth
            idx = (cols[:, None] * height + np.arange(height)[None, :]).ravel()
            c, s = math.cos(yaw[b]), math.sin(yaw[b])
            Rb = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])        # world -> box
            ob = Rb @ (o - cz[b])
            db = d[idx] @ Rb.T
            inv = 1.0 / db
            t1 = (-half[b] - ob) * inv
            t2 = (half[b] - ob) * inv
            tmin = np.nanmax(np.minimum(t1, t2), axis=1)
            tmax = np.nanmin(np.maximum(t1, t2), axis=1)
            ok = (tmax >= tmin) & (tmin > 0)
            t = np.where(ok, tmin, np.inf)
            hit = t < t_best[idx]
            t_best[idx] = np.where(hit, t, t_best[idx])
            label[idx] = np.where(hit, blabel[b], label[idx])

    keep = np.isfinite(t_best) & (t_best < max_range) & (t_best > 0.5)
    r = t_best[keep] * (1.0 + (noise * rng.standard_normal(keep.sum())) / np.maximum(t_best[keep], 1.0))
    pts = np.ones((r.shape[0], 4), dtype=np.float32)
    pts[:, :3] = (d_local[keep] * r[:, None]).astype(np.float32)
    if semantics:
        labels = label[keep].astype(np.float32)
        probs = np.random.default_rng(42 + k).uniform(0.6, 1.0, r.shape[0]).astype(np.float32)
    else:
        labels = np.zeros(r.shape[0], dtype=np.float32)
        probs = np.zeros(r.shape[0], dtype=np.float32)
    return pts, labels, probs, T
