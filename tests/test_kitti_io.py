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
    lab = rng.choice([0, 10, 40, 50, 252, 259], 1000).astype("<u4") | (rng.integers(0, 100, 1000).astype("<u4") << 16)
    lab.tofile(root / "labels" / "000000.label")
    (root / "calib.txt").write_text("P0: " + " ".join(["1", "0", "0", "0", "0", "1", "0", "0", "0", "0", "1", "0"]) +
                                    "\nTr: 0 -1 0 0.1 0 0 -1 0.2 1 0 0 0.3\n")
    seq = kitti.Sequence(str(root))
    assert len(seq) == 1
    pts, labels, probs = seq[0]
    assert pts.dtype == np.float32 and pts.shape == (1000, 4)
    assert np.array_equal(pts[:, :3], raw[:, :3]) and (pts[:, 3] == 1).all()  # remission dropped, w = 1
    assert set(np.unique(labels)) <= {0.0, 10.0, 40.0, 50.0, 20.0}            # instance bits stripped, moving -> static
    assert (probs == 1).all()
    Tr = seq.calib["Tr"]
    assert Tr.shape == (4, 4) and Tr[3, 3] == 1 and Tr[0, 3] == 0.1
    rows = kitti.poses_to_camera_frame([np.eye(4)], Tr)
    assert np.allclose(rows[0], np.eye(4)[:3].reshape(12))
