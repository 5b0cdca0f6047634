#!/usr/bin/env python
"""Generates tests/golden/gl_gn_steps_900x64.npz: THE ACCEPTANCE LINE AS A FIXTURE.  For every Gauss-Newton iteration of
scans 1 .. 4 (64 x 900, ten iterations each) of the teacher-forced run of tests/test_gl_controls.py: the pose before the
iteration (the oracle's history) and the pose after ONE step of the reference's Frame2Model_jacobians.{vert,geom,frag}
executed by a real OpenGL (Mesa llvmpipe) on the same frames -- the reference's shader text unchanged, its asin / acos /
atan taken from the specified functions (oracle/glref.py::DETMATH_PRELUDE; GLSL leaves their accuracy to the
implementation and llvmpipe's asin is 3.9e-4 rad off) -- and, beside it (pose_after_gl_driver), the same step with the
GL implementation's OWN transcendentals: the reference path as shipped, the check that depends on nothing this repository
specifies (round-5 advisor).  tests/test_gpu_gl_golden.py runs the same teacher-forced
minimisations through the HIP path on the GPU box (no Mesa, no /root/reference there) and asserts, iteration by iteration,
north_star's "pose delta within 1e-4 m / 1e-5 rad per ICP iteration" against the poses stored here.
Run from the repo root:   python tests/golden/make_gl_gn_steps_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import glref, pyoracle, pyref  # noqa: E402
from semantic_suma_amd import synth  # noqa: E402
from semantic_suma_amd.types import params_with_size  # noqa: E402

W, H, N, ITER = 900, 64, 5, 10
p = params_with_size(W, H)
op = pyoracle.OraclePipeline(p, threads=8)
with glref.transcendentals("detmath"):
    k6 = glref.Jacobians(p)
with glref.transcendentals("driver"):
    k6_driver = glref.Jacobians(p)  # the reference GL path AS SHIPPED: the GL implementation's own asin / acos / atan
before, after_gl, after_gl_driver, after_oracle = [], [], [], []
for k in range(N):
    pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W, height=H)
    if k >= 1:
        ora = op.ctx
        cur = ora.preprocess(pts, lab, prob, k, ora.frame())
        pose32 = op.pose().astype(np.float32)
        out = ora.frame(model=True)
        ct = float(np.float32((1.0 - k / 10.0) * math.log(0.1 / 0.9) + np.float32(k / 10.0) * np.float32(p.confidence_threshold)))
        ora.map_render(pose32, pose32, ct, out)
        model = ora.map_frame(1)
        ora.set_params(params_with_size(W, H, max_iterations=ITER, stopping_threshold=0.0, delta=0.0))
        _, hist, _ = ora.minimize(cur, model, op.last_increment(), history_cap=ITER + 1)
        ora.set_params(p)
        cm, mm = [cur.map(m).copy() for m in range(3)], [model.map(m).copy() for m in range(3)]
        for it in range(ITER):
            b = k6.run(cm, mm, hist[it], it)
            dx = np.linalg.solve(b[:36].reshape(6, 6).astype(np.float64), -b[36:42].astype(np.float64))  # LieGaussNewton.cpp:60
            before.append(hist[it])
            after_gl.append(pyref.se3_exp(dx) @ hist[it])
            b = k6_driver.run(cm, mm, hist[it], it)
            dx = np.linalg.solve(b[:36].reshape(6, 6).astype(np.float64), -b[36:42].astype(np.float64))
            after_gl_driver.append(pyref.se3_exp(dx) @ hist[it])
            after_oracle.append(hist[it + 1])
    op.process_scan(pts, lab, prob, fixed_iterations=ITER)
out = os.path.join(ROOT, "tests", "golden", f"gl_gn_steps_{W}x{H}.npz")
np.savez_compressed(out, W=W, H=H, scans=N, iterations=ITER, pose_before=np.array(before), pose_after_gl=np.array(after_gl),
                    pose_after_gl_driver=np.array(after_gl_driver), pose_after_oracle=np.array(after_oracle), gl_version=glref.limits()["version"],
                    gl_renderer=glref.limits()["renderer"], transcendentals="include/suma_detmath.h (GLSL prelude)")
d = np.array([np.linalg.norm((np.linalg.inv(a) @ b)[:3, 3]) for a, b in zip(after_gl, after_oracle)])
dd = np.array([np.linalg.norm((np.linalg.inv(a) @ b)[:3, 3]) for a, b in zip(after_gl_driver, after_oracle)])
print(f"{out}: {len(before)} Gauss-Newton steps; GL vs oracle: worst {d.max():.2e} m under specified transcendentals, "
      f"{dd.max():.2e} m (median {np.median(dd):.2e}) with the driver's; {glref.limits()['renderer']}")
