"""KITTI readers (semantic_suma_amd/kitti.py) on files written in the dataset's formats."""
import os

import numpy as np

from semantic_suma_amd import kitti


def test_sequence_roundtrip(tmp_path):
    root = tmp_path / "sequences" / "00"
    (root / "velodyne").mkdir(parents=True)
    (root / "labels").mkdir()
    rng = np.random.default_rng(0)
    raw = rng.normal(size=(1000, 4)).astype("<f4")
    raw.tofile(root / "velodyne" / "000000.bin")
    lab = rng.choice([0, 1, 10, 13, 16, 40, 50, 52, 60, 99, 252, 256, 257, 259], 1000).astype("<u4") | (rng.integers(0, 100, 1000).astype("<u4") << 16)
    lab.tofile(root / "labels" / "000000.label")
    (root / "calib.txt").write_text("P0: " + " ".join(["1", "0", "0", "0", "0", "1", "0", "0", "0", "0", "1", "0"]) +
                                    "\nTr: 0 -1 0 0.1 0 0 -1 0.2 1 0 0 0.3\n")
    seq = kitti.Sequence(str(root))
    assert len(seq) == 1
    pts, labels, probs = seq[0]
    assert pts.dtype == np.float32 and pts.shape == (1000, 4)
    assert np.array_equal(pts[:, :3], raw[:, :3]) and (pts[:, 3] == 1).all()  # remission dropped, w = 1
    # instance bits stripped; learning_map then learning_map_inv, i.e. what RangeNet++ reports (KITTIReader.cpp:189-200)
    assert set(np.unique(labels)) == {0.0, 10.0, 20.0, 40.0, 50.0}
    want = {0: 0, 1: 0, 10: 10, 13: 20, 16: 20, 40: 40, 50: 50, 52: 0, 60: 40, 99: 0, 252: 10, 256: 20, 257: 20, 259: 20}
    assert all(labels[i] == want[int(lab[i] & 0xFFFF)] for i in range(1000))
    assert kitti.remap_labels(np.array([70, 71, 72, 80, 81, 253, 254, 255, 258, 4000])).tolist() == \
        [70, 71, 72, 80, 81, 31, 30, 32, 18, 0]
    assert (probs == 1).all()
    Tr = seq.calib["Tr"]
    assert Tr.shape == (4, 4) and Tr[3, 3] == 1 and Tr[0, 3] == 0.1
    rows = kitti.poses_to_camera_frame([np.eye(4)], Tr)
    assert np.allclose(rows[0], np.eye(4)[:3].reshape(12))


def test_devkit_odometry_errors_known_answers():
    """the KITTI odometry devkit metric as the reference vendors it (src/util/kitti_utils.cpp:108-191), on trajectories
    with known errors: none; a 1 % scale error (t_err = 0.01 exactly on a straight line); a constant yaw drift per metre"""
    n = 900
    gt = np.tile(np.eye(4), (n, 1, 1))
    gt[:, 0, 3] = np.arange(n) * 1.0                       # 1 m per frame, straight
    e = kitti.odometry_errors(gt, gt)
    assert e["t_err"] == 0.0 and e["r_err"] == 0.0 and e["segments"] > 100 and set(e["per_length"]) == set(kitti.SEGMENT_LENGTHS)
    scaled = gt.copy()
    scaled[:, 0, 3] *= 1.01
    e = kitti.odometry_errors(gt, scaled)
    assert abs(e["t_err"] - 0.01) < 1e-3 and e["r_err"] < 1e-12   # the segment ends one frame behind the nominal length
    # heading drifts by 1e-4 rad per metre: r_err = 1e-4 rad/m on every segment
    drift = gt.copy()
    for k in range(n):
        a = 1e-4 * k
        drift[k, :2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    e = kitti.odometry_errors(gt, drift)
    assert abs(e["r_err"] - 1e-4) < 2e-6
    assert kitti.odometry_errors(gt[:50], gt[:50]) is None     # 49 m: no segment fits
    # poses file round trip through the camera frame
    Tr = np.array([[0, -1, 0, 0.1], [0, 0, -1, 0.2], [1, 0, 0, 0.3], [0, 0, 0, 1.0]])
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        np.savetxt(os.path.join(d, "00.txt"), kitti.poses_to_camera_frame(drift[:20], Tr))
        back = kitti.read_poses(os.path.join(d, "00.txt"), Tr)
    assert np.allclose(back, drift[:20], atol=1e-9)
