#!/usr/bin/env python
"""Generates tests/golden/*.npz from the CPU oracle (oracle/).

The reference cannot be built or run (OpenGL + glow + Eigen + gtsam + Qt) and ships no
golden vectors, so these fixtures do NOT pin the oracle to the reference (that is the job of oracle/_ref +
tests/test_ref_shaders.py, which compile the reference's own shaders); they pin
the oracle -- and, on the GPU, the HIP path -- against regressions, on a fixed seeded input.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from semantic_suma_amd import synth  # noqa: E402
from semantic_suma_amd.types import params_with_size  # noqa: E402

W, H, N_SCANS = 360, 32, 4


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    p = params_with_size(W, H)
    pipe = pyoracle.OraclePipeline(p)
    out = {"W": W, "H": H, "n_scans": N_SCANS}
    for k in range(N_SCANS):
        pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W, height=H)
        out[f"pts{k}"], out[f"lab{k}"], out[f"prob{k}"] = pts, lab, prob
        pipe.process_scan(pts, lab, prob, fixed_iterations=10)
        f = pipe.frame(0)
        out[f"sha_vertex{k}"], out[f"sha_normal{k}"], out[f"sha_semantic{k}"] = sha(f.vertex), sha(f.normal), sha(f.semantic)
        out[f"pose{k}"] = pipe.pose()
        s = pipe.ctx.map_surfels()
        out[f"map_size{k}"] = len(s)
        out[f"sha_map{k}"] = sha(s)
        out[f"counts{k}"] = np.array(pipe.ctx.map_counts())
        st = pipe.last_stats()
        out[f"stats{k}"] = np.array([st.valid, st.outlier, st.inlier, st.invalid, st.iterations])
        out[f"stats_f{k}"] = np.array([st.error, st.inlier_residual])
        out[f"sha_model{k}"] = sha(pipe.frame(2).vertex)
    # one K6 evaluation with all raw fixed-point words
    ora = pyoracle.Oracle(p)
    f0 = ora.preprocess(out["pts0"], out["lab0"], out["prob0"], 20, ora.frame())
    f1 = ora.preprocess(out["pts1"], out["lab1"], out["prob1"], 21, ora.frame())
    T = np.eye(4)
    T[0, 3] = 1.0
    F, acc, JtJ, Jtr, st = ora.jacobian_products(f1, f0, T, 0)
    out["k6_acc"], out["k6_T"] = acc, T
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_360x32.npz"), **out)
    print("wrote pipeline_360x32.npz:", {k: out[k] for k in ("map_size0", "map_size3")}, "x =", out["pose3"][0, 3])


if __name__ == "__main__":
    main()
