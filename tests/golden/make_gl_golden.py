#!/usr/bin/env python
"""Generates tests/golden/gl_render_450x32.npz: SurfelMap::render_active as A REAL OPENGL IMPLEMENTATION executes it --
the reference's own render_surfels.{vert,geom,frag} (read from /root/reference/src/shader), run by Mesa llvmpipe through
oracle/glref.py with the GL state of SurfelMap.cpp:1023-1069.  The map (surfels + pose table) comes from a short oracle
run and is stored, so that the GPU suite can render the same map with the HIP path and compare (tests/test_gpu_gl_golden.py);
/root/reference and Mesa are only needed here, at generation time.  Run from the repo root:
    python tests/golden/make_gl_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import glref, pyoracle  # noqa: E402
from semantic_suma_amd import synth  # noqa: E402
from semantic_suma_amd.types import params_with_size  # noqa: E402

W, H, N = 450, 32, 8
p = params_with_size(W, H)
op = pyoracle.OraclePipeline(p)
for k in range(N):
    pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W, height=H)
    op.process_scan(pts, lab, prob, fixed_iterations=10)
ctx = op.ctx
surfels, poses, ts = ctx.map_surfels(), ctx.map_poses(N + 1).reshape(-1, 16), ctx.map_timestamp()
pose = op.pose().astype(np.float32).astype(np.float64)
ct = 0.0
R = glref.SurfelRenderer(p)
new = R.render(surfels, poses, pose, ct, ts - 100, False)    # render_active: "new" surfels (SurfelMap.cpp:1040-1052)
out = os.path.join(ROOT, "tests", "golden", f"gl_render_{W}x{H}.npz")
np.savez_compressed(out, W=W, H=H, timestamp=ts, pose=pose, conf_threshold=ct, surfels=surfels, poses=poses,
                    gl_vertex=new[0], gl_normal=new[1], gl_semantic=new[2], gl_version=glref.limits()["version"],
                    gl_renderer=glref.limits()["renderer"])
print(f"{out}: {os.path.getsize(out) // 1024} KB; {surfels.shape[0]} surfels, {int((new[0][..., 3] > 0.5).sum())} texels rendered by "
      f"{glref.limits()['renderer']}")
