#!/usr/bin/env python
"""Generates tests/golden/gl_pipeline_900x64.npz: the trajectory and the map sizes of eight scans processed by
SurfelMapping::processScan AS A REAL OPENGL IMPLEMENTATION EXECUTES IT -- every pass K1-K12 is the reference's GLSL
(/root/reference/src/shader) run by Mesa llvmpipe, the host code between the passes restated in numpy
(oracle/glpipeline.py).  The scans are the seeded synthetic ones (semantic_suma_amd/synth.py), so the GPU suite can
run the same scans through the HIP pipeline and compare (tests/test_gpu_gl_golden.py) without Mesa or /root/reference.
Run from the repo root:   python tests/golden/make_gl_pipeline_golden.py [bench]     (bench: 64 x 2048, six scans)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import glpipeline, glref  # noqa: E402
from semantic_suma_amd import synth  # noqa: E402
from semantic_suma_amd.types import params_with_size  # noqa: E402

W, H, N, ITER = (2048, 64, 6, 10) if (len(sys.argv) > 1 and sys.argv[1] == "bench") else (900, 64, 8, 10)
p = params_with_size(W, H)
g = glpipeline.GLPipeline(p)
poses, incs, counts = [], [], []
for k in range(N):
    pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W, height=H)
    g.process_scan(pts, lab, prob, ITER)
    poses.append(g.current_pose.copy())
    incs.append(g.last_increment.copy())
    counts.append([g.counts[c] for c in ("map", "updated", "new", "integrated", "index")])
out = os.path.join(ROOT, "tests", "golden", f"gl_pipeline_{W}x{H}.npz")
np.savez_compressed(out, W=W, H=H, scans=N, iterations=ITER, poses=np.array(poses), increments=np.array(incs),
                    counts=np.array(counts, dtype=np.int64), count_names="map updated new integrated index",
                    gl_version=glref.limits()["version"], gl_renderer=glref.limits()["renderer"])
print(f"{out}: {N} scans, x = {poses[-1][0, 3]:.3f} m, map {counts[-1][0]} surfels; {glref.limits()['renderer']}")
