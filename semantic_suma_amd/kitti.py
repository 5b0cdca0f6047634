"""KITTI odometry input: the formats the reference's readers consume, as numpy.

  velodyne/NNNNNN.bin   N x 4 float32 little-endian (x, y, z, remission) -- reference
                        src/io/KITTIReader.cpp:140-167 turns each row into rv::Point3f (x, y, z, 1)
  labels/NNNNNN.label   N x uint32 SemanticKITTI labels (lower 16 bits = class id); stands in for the
                        RangeNet++ inference of src/io/KITTIReader.cpp:175-200, which produces a class id
                        per point (labels_float) and its softmax probability (labels_prob)
  calib.txt / poses     src/util/kitti_utils.cpp:32-61 (P0..P3, Tr); poses are written in the camera frame
                        (src/visualizer/VisualizerWindow.cpp:848-872): pose_cam = Tr * pose_velo * Tr^-1

No dataset ships with this repository and the build machines have no network: `bench.py` uses these
readers only when SUMA_KITTI_DIR points at `sequences/XX`.
"""
from __future__ import annotations

import glob
import os

import numpy as np


def read_velodyne(path: str) -> np.ndarray:
    """-> points[N, 4] float32 (x, y, z, 1) as rv::Point3f"""
    raw = np.fromfile(path, dtype="<f4")
    if raw.size % 4:
        raise ValueError(f"{path}: size is not a multiple of 4 floats")
    pts = raw.reshape(-1, 4).copy()
    pts[:, 3] = 1.0
    return pts


# SemanticKITTI learning_map (raw id -> one of RangeNet++'s 20 training classes) and learning_map_inv (training
# class -> the raw id it is reported as).  The reference only ever sees learning_map_inv values: KITTIReader.cpp:189-200
# assigns labels[i] = label_map_[argmax_j], j < 20.
LEARNING_MAP = {0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10, 48: 11,
                49: 12, 50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0, 252: 1, 253: 7,
                254: 6, 255: 8, 256: 5, 257: 5, 258: 4, 259: 5}
LEARNING_MAP_INV = {0: 0, 1: 10, 2: 11, 3: 15, 4: 18, 5: 20, 6: 30, 7: 31, 8: 32, 9: 40, 10: 44, 11: 48, 12: 49, 13: 50,
                    14: 51, 15: 70, 16: 71, 17: 72, 18: 80, 19: 81}


def remap_labels(raw_ids: np.ndarray) -> np.ndarray:
    """raw SemanticKITTI ids -> the ids RangeNet++ would report: learning_map followed by learning_map_inv
    (e.g. moving-car 252 -> 10, bus 13 / on-rails 16 / moving-bus 257 -> other-vehicle 20, lane-marking 60 -> road 40,
    outlier 1 / other-structure 52 / other-object 99 -> unlabeled 0).  Unknown ids map to 0."""
    lut = np.zeros(260, dtype=np.float32)
    for raw, train in LEARNING_MAP.items():
        lut[raw] = LEARNING_MAP_INV[train]
    ids = np.asarray(raw_ids, dtype=np.int64)
    return lut[np.where((ids >= 0) & (ids < 260), ids, 0)]


def read_labels(path: str, n_points: int, prob: float = 1.0):
    """SemanticKITTI .label -> (labels_float[N], labels_prob[N]) as the reference's reader would deliver them from
    RangeNet++ (KITTIReader.cpp:175-200); ground-truth labels carry probability `prob`"""
    raw = np.fromfile(path, dtype="<u4")
    if raw.size != n_points:
        raise ValueError(f"{path}: {raw.size} labels for {n_points} points")
    return remap_labels(raw & 0xFFFF), np.full(n_points, prob, dtype=np.float32)


def read_calib(path: str) -> dict:
    """kitti_utils.cpp:32-61: 'P0: ...' lines of 12 floats -> 4x4 matrices (last row 0 0 0 1)"""
    out = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            key, vals = line.split(":", 1)
            v = np.array(vals.split(), dtype=np.float64)
            if v.size == 12:
                M = np.eye(4)
                M[:3, :4] = v.reshape(3, 4)
                out[key.strip()] = M
    return out


def poses_to_camera_frame(poses_velo, Tr: np.ndarray) -> np.ndarray:
    """VisualizerWindow.cpp:848-872: pose_cam = Tr * pose_velo * Tr^-1, flattened to the 12-value KITTI rows"""
    Tinv = np.linalg.inv(Tr)
    return np.stack([(Tr @ np.asarray(P) @ Tinv)[:3, :4].reshape(12) for P in poses_velo])


class Sequence:
    """one `sequences/XX` directory"""

    def __init__(self, root: str):
        self.root = root
        self.scans = sorted(glob.glob(os.path.join(root, "velodyne", "*.bin")))
        if not self.scans:
            raise FileNotFoundError(f"no velodyne/*.bin under {root}")
        calib = os.path.join(root, "calib.txt")
        self.calib = read_calib(calib) if os.path.exists(calib) else {}

    def __len__(self):
        return len(self.scans)

    def __getitem__(self, k: int):
        pts = read_velodyne(self.scans[k])
        lab_path = os.path.join(self.root, "labels", os.path.basename(self.scans[k]).replace(".bin", ".label"))
        if os.path.exists(lab_path):
            labels, probs = read_labels(lab_path, pts.shape[0])
        else:
            labels = np.zeros(pts.shape[0], np.float32)
            probs = np.zeros(pts.shape[0], np.float32)
        return pts, labels, probs


# ---- KITTI odometry devkit error metric (src/util/kitti_utils.cpp:108-191: trajectoryDistances,
#      lastFrameFromSegmentLength, rotationError, translationError, calcSequenceErrors) ----
SEGMENT_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)


def odometry_errors(poses_gt, poses_est, lengths=SEGMENT_LENGTHS, step: int = 10):
    """Relative pose error over all sub-sequences of the devkit's lengths, every `step` frames (kitti_utils.cpp:153-191):
    returns dict(t_err = mean translation error per metre, r_err = mean rotation error in rad per metre, segments = number
    of (start, length) pairs that fit, per_length = {length: (t_err, r_err, count)}).  Distances are measured along the
    ground truth; a pair contributes only if the trajectory is long enough.  None if no segment fits."""
    G = np.asarray(poses_gt, dtype=np.float64).reshape(-1, 4, 4)
    E = np.asarray(poses_est, dtype=np.float64).reshape(-1, 4, 4)
    n = min(len(G), len(E))
    if n < 2:
        return None
    G, E = G[:n], E[:n]
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(G[1:, :3, 3] - G[:-1, :3, 3], axis=1))])
    per = {}
    for first in range(0, n, step):
        for length in lengths:
            later = np.nonzero(dist[first:] > dist[first] + length)[0]  # lastFrameFromSegmentLength
            if later.size == 0:
                continue
            last = first + int(later[0])
            d_gt = np.linalg.inv(G[first]) @ G[last]
            d_est = np.linalg.inv(E[first]) @ E[last]
            err = np.linalg.inv(d_est) @ d_gt
            r = float(np.arccos(max(-1.0, min(1.0, 0.5 * (np.trace(err[:3, :3]) - 1.0)))))
            t = float(np.linalg.norm(err[:3, 3]))
            per.setdefault(length, []).append((t / length, r / length))
    if not per:
        return None
    allv = np.array([v for vs in per.values() for v in vs])
    return dict(t_err=float(allv[:, 0].mean()), r_err=float(allv[:, 1].mean()), segments=int(len(allv)),
                per_length={int(k): (float(np.mean([v[0] for v in vs])), float(np.mean([v[1] for v in vs])), len(vs))
                            for k, vs in sorted(per.items())})


def read_poses(path: str, Tr: np.ndarray = None) -> np.ndarray:
    """poses/XX.txt of the odometry benchmark: 12 values per line, camera frame; with the calibration's Tr they are
    brought into the velodyne frame (the inverse of poses_to_camera_frame)"""
    rows = np.loadtxt(path).reshape(-1, 12)
    P = np.tile(np.eye(4), (len(rows), 1, 1))
    P[:, :3, :4] = rows.reshape(-1, 3, 4)
    if Tr is not None:
        Ti = np.linalg.inv(Tr)
        P = np.stack([Ti @ p @ Tr for p in P])
    return P
