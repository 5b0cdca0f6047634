"""TEST INFRASTRUCTURE.  SurfelMapping::processScan with the reference's OWN shaders executed by a real OpenGL.

The closest thing to "running the reference" these machines allow: every GL pass of the hot path (K1-K12) is the
reference's GLSL text from /root/reference/src/shader, compiled and executed by Mesa llvmpipe through oracle/glref.py;
the host code between the passes -- what SurfelMapping.cpp, SurfelMap.cpp, Frame2Model.cpp and LieGaussNewton.cpp do
with glow, Eigen and the CPU -- is restated here in numpy, each step citing the lines it follows, including the submap
window (updateActiveSubmaps: tiles pushed for extraction, extract_surfels through transform feedback, parked tiles
appended when they come back) and the frame-to-frame fallback minimisation after a track loss.  Loop closures and the
pose graph are outside.

Used by tests/test_gl_pipeline.py only: it puts the end-to-end acceptance criterion of the task -- poses within
1e-4 m / 1e-5 rad per ICP iteration of the reference's OpenGL path on identical scans -- on a real GL implementation
instead of on the builder's restatement alone.  Nothing in the product path, bench.py or smoke() imports this file.
"""
import math

import numpy as np

from . import glref as gl
from . import pyref

f32, f64 = np.float32, np.float64


def rigid_inverse_f32(pose):
    """R^T, -R^T t in float: what the HIP path and the oracle hand to the shaders as inv_pose (the
    reference calls Eigen's general Matrix4f::inverse(), equal on rigid input to float rounding)"""
    return pyref.rigid_inverse_f32(pose)


class GLPipeline:
    def __init__(self, params):
        p = self.p = params
        self.W, self.H = p.data_width, p.data_height
        self.prelude = gl._prelude[0]  # the transcendentals the programs are built with (gl.transcendentals), kept for the lazy ones
        self.ref = pyref.Ref(p)  # the uniform tables (values as the reference's host code computes them) + SE3::exp
        self.k1, self.k23 = gl.VertexMap(p), gl.NormalsLabels(p)
        self.k4, self.k6 = gl.SurfelRenderer(p), gl.Jacobians(p)
        self.k7, self.k8 = gl.IndexMap(p), gl.RadiusConfidence(p)
        self.k9, self.k10 = gl.SurfelUpdate(p), gl.SurfelGenerate(p)
        self.k11 = gl.SurfelFilter("copy_surfels.vert")
        self.k12 = gl.SurfelFilter("extract_surfels.vert")
        self.origin = [0, 0]       # submap_origin_
        self.extraction = []       # extraction_buffer_ (tiles waiting to be parked)
        self.cache = {}            # submapCache_(i, j).surfels
        self.extractions = 0
        self.timestamp = 0
        self.surfels = np.zeros((0, 16), dtype=f32)
        eye_cm = np.eye(4, dtype=f32).reshape(-1)
        self.poses = np.tile(eye_cm, (int(p.max_poses), 1))  # poses_, column-major 4x4 floats (SurfelMap.h:205)
        self.current_pose = np.eye(4)
        self.last_increment = np.eye(4)
        self.frame = None
        self.last_frame = None
        self.k6_fallback = None
        self.track_loss = 0
        self.counts = {}
        p_unstable = f32(0.1)  # SurfelMapping.cpp:108-109
        self.log_unstable = f32(math.log(float(p_unstable / (f32(1.0) - p_unstable))))

    def center(self, i, j):
        """submapIndex2center, SurfelMap.cpp:704-706"""
        return (f32(2.0 * i * self.p.submap_extent), f32(2.0 * j * self.p.submap_extent))

    def update_active_submaps(self, pose32, poses):
        """SurfelMap::updateActiveSubmaps (SurfelMap.cpp:744-824) and extractSurfels (:708-742)"""
        p = self.p
        dim, ext = int(p.submap_dimension), f32(p.submap_extent)
        cx, cy = self.center(*self.origin)
        changex, changey = f32(pose32[0, 3]) - cx, f32(pose32[1, 3]) - cy
        limit = f32(1.1) * ext

        def append(i, j):  # :775-780, 801-806: the parked tile comes back behind the active surfels
            tile = self.cache.get((i, j))
            if tile is not None and tile.shape[0]:
                room = int(p.max_surfels) - self.surfels.shape[0]
                self.surfels = np.concatenate([self.surfels, tile[:max(room, 0)]])

        if abs(changex) > limit:
            d = -1 if changex < 0 else 1
            self.extraction += [(self.origin[0] - d * dim, self.origin[1] + k) for k in range(-dim, dim + 1)]
            self.origin[0] += d
            for k in range(-dim, dim + 1):
                append(self.origin[0] + d * dim, self.origin[1] + k)
        if abs(changey) > limit:
            d = -1 if changey < 0 else 1
            self.extraction += [(self.origin[0] + r, self.origin[1] - d * dim) for r in range(-dim, dim + 1)]
            self.origin[1] += d
            for r in range(-dim, dim + 1):
                append(self.origin[0] + r, self.origin[1] + d * dim)
        while self.extraction:  # extractSurfels(partial_extraction_): tiles are taken from the back
            i, j = self.extraction.pop()
            tile = self.k12.run([self.surfels], poses, self.center(i, j), ext)
            self.cache[(i, j)] = tile[:500000]  # extractBuffer_.reserve(500000), :279
            self.extractions += 1
            self.last_extraction = (i, j)
            if p.partial_extraction:
                break

    def conf_threshold(self):
        """SurfelMapping::getConfidenceThreshold, SurfelMapping.cpp:333-340 (time_init = 10)"""
        ct = f32(self.p.confidence_threshold)
        if self.timestamp < 10:
            alpha = f32(self.timestamp) / f32(10)
            ct = f32((1.0 - float(alpha)) * float(self.log_unstable) + float(alpha * f32(self.p.confidence_threshold)))
        return float(ct)

    def render_new(self, pose32):
        """the NEW frame of SurfelMap::render(pose, pose, ct) (SurfelMap.cpp:847-1021): active surfels, timestamp
        threshold timestamp_ - 100 (:873)"""
        return self.k4.render(self.surfels, self.poses[: self.timestamp + 1], pose32, self.conf_threshold(),
                              self.timestamp - 100, False)

    def minimize(self, cur, model, T0, iterations, history=None):
        """LieGaussNewton::minimize with a fixed number of steps (LieGaussNewton.cpp:13-79; the stopping tests of :64-66
        are off in the fixed-iteration runs compared here): JtJ.ldlt().solve(-Jtf) in double, pose = exp(delta) * pose"""
        Tk = np.array(T0, dtype=f64)
        for it in range(iterations):
            if history is not None:
                history.append(Tk.copy())
            b = self.k6.run(cur, model, Tk, it)  # Frame2Model::jacobianProducts, iteration_ counts the calls
            JtJ, Jtr = b[:36].reshape(6, 6).astype(f64), b[36:42].astype(f64)  # Frame2Model.cpp:214-227
            dx = np.linalg.solve(JtJ, -Jtr)
            Tk = pyref.se3_exp(dx) @ Tk
        if history is not None:
            history.append(Tk.copy())
        return Tk

    def process_scan(self, points, labels, probs, iterations):
        p, t = self.p, self.timestamp
        # initialize() + preprocess(), SurfelMapping.cpp:181-187, 323-358
        gv, gs = self.k1.run(points, labels, probs, t, int(p.label_offset), int(p.prob_offset))
        gn, _, gr = self.k23.run(gv, gs)
        self.last_frame = self.frame  # initialize(), SurfelMapping.cpp:323-331: the frames swap
        frame = self.frame = (gv, gn, gr)
        pose32 = self.current_pose.astype(f32)
        if t > 0:  # updatePose(), :372-476
            model = self.render_new(pose32)
            increment = self.minimize(frame, model, self.last_increment, iterations)
            delta = np.linalg.inv(self.last_increment) @ increment
            t_err = float(np.linalg.norm(delta[:3, 3]))
            r_err = math.acos(max(-1.0, min(1.0, 0.5 * (np.trace(delta[:3, :3]) - 1.0))))
            if t > 1 and (f32(t_err) > 0.4 or f32(r_err) > 0.1) and p.fallback_mode:
                # track loss, SurfelMapping.cpp:438-449: frame-to-frame with the fallback gates, same start
                self.track_loss += 1
                if self.k6_fallback is None:
                    import copy
                    pf = copy.copy(p)
                    pf.icp_max_distance, pf.icp_max_angle = p.fallback_max_distance, p.fallback_max_angle
                    saved, gl._prelude[0] = gl._prelude[0], self.prelude
                    try:
                        self.k6_fallback = gl.Jacobians(pf)
                    finally:
                        gl._prelude[0] = saved
                k6, self.k6 = self.k6, self.k6_fallback
                increment = self.minimize(frame, self.last_frame, self.last_increment, iterations)
                self.k6 = k6
            self.current_pose = self.current_pose @ increment
            self.last_increment = increment
        # updateMap() -> SurfelMap::update(pose, frame), SurfelMap.cpp:492-584
        pose32 = self.current_pose.astype(f32)
        inv32 = rigid_inverse_f32(pose32)
        self.poses[t] = np.ascontiguousarray(pose32.T).reshape(-1)  # poses_[timestamp_] = pose, :494
        poses = self.poses[: t + 2]
        idx, _ = self.k7.run(self.surfels, poses, pose32, inv32) if self.surfels.shape[0] else (np.zeros((self.H, self.W), np.uint32), None)
        uni = dict(fov_up=float(abs(f32(p.data_fov_up))), fov_down=float(abs(f32(p.data_fov_down))), min_depth=float(f32(p.min_depth)),
                   max_depth=float(f32(p.max_depth)), pixel_size=float(self.ref.pixel_size), confidence_mode=int(p.confidence_mode),
                   min_radius=float(f32(p.min_radius)), max_radius=float(f32(p.max_radius)),
                   angle_thresh=float(f32(math.cos(float(f32(float(f32(p.max_angle)) * math.pi / 180.0))))))
        rc, _ = self.k8.run(uni, gv, gn)
        if self.surfels.shape[0]:
            upd, mask = self.k9.run(self.ref.update_uniforms(pose32, t), self.surfels, poses, frame, rc, idx.astype(f32))
        else:
            upd, mask = np.zeros((0, 16), f32), np.zeros((self.H, self.W), f32)
        mask4 = np.zeros((self.H, self.W, 4), f32)
        mask4[..., 0] = mask > 0.5
        new = self.k10.run(self.ref.generate_uniforms(pose32, t), frame, rc, mask4)
        extent = f32(2.0) * f32(p.submap_dimension) * f32(p.submap_extent) + f32(p.submap_extent)  # copySurfels, :667-677
        if p.partial_extraction and self.extraction:
            extent = extent + f32(2.0) * f32(p.submap_extent)  # :674-677
        self.surfels = self.k11.run([upd, new], poses, self.center(*self.origin), extent)
        self.update_active_submaps(pose32, poses)
        self.counts = dict(updated=int(upd.shape[0]), new=int(new.shape[0]), map=int(self.surfels.shape[0]),
                           integrated=int((mask > 0.5).sum()), index=int((idx > 0).sum()))
        self.timestamp += 1
