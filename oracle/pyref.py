"""TEST INFRASTRUCTURE -- ctypes front end of oracle/_ref/libsuma_ref*.so: the reference's OWN GLSL shaders
(/root/reference/src/shader, compiled to C++ by oracle/ref_build.py) and src/core/lie_algebra.cpp, executed on the CPU.

Only tests/ import this module.  What is restated HERE is the host side of each draw call -- which uniform gets which
value and which texture / sampler state is bound -- with the reference line it comes from; the per-vertex /
per-fragment arithmetic is the reference's source text.

The libraries are built in this container (where /root/reference exists) and travel with the tree; on a machine
without them `available()` is False and the tests skip.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

from semantic_suma_amd.types import SURFEL_DTYPE, SumaParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
NEAREST, LINEAR, CLAMP_TO_EDGE = 0, 1, 2


def build(force=False):
    cmd = [sys.executable, os.path.join(_HERE, "ref_build.py")] + (["--force"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)


def _path(variant):
    return os.path.join(_HERE, "_ref", "libsuma_ref.so" if not variant else f"libsuma_ref_{variant}.so")


def available(variant=""):
    if not os.path.exists(_path(variant)) and os.path.isdir("/root/reference/src/shader"):
        build()
    return os.path.exists(_path(variant))


def lib(variant=""):
    if variant in _LIBS:
        return _LIBS[variant]
    if not available(variant):
        raise RuntimeError("oracle/_ref is not built (needs /root/reference; run python oracle/ref_build.py)")
    L = C.CDLL(_path(variant))
    vp, i32, u32, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_float
    L.ref_set_uniform.argtypes = [C.c_char_p, C.c_char_p, vp]
    L.ref_bind_texture.argtypes = [C.c_char_p, C.c_char_p, vp, i32, i32, i32, i32]
    L.ref_bind_buffer.argtypes = [C.c_char_p, C.c_char_p, vp, i32]
    L.ref_draw_vertexmap.argtypes = [vp, vp, vp, u32, i32, i32, vp, vp]
    L.ref_draw_vertexmap_blend.argtypes = [vp, vp, vp, u32, i32, i32, vp, vp]
    for n in ("ref_pass_normalmap",):
        getattr(L, n).argtypes = [i32, i32, vp, vp]
    for n in ("ref_pass_floodfill", "ref_pass_avg_vertexmap", "ref_pass_bilateral", "ref_draw_radius_conf"):
        getattr(L, n).argtypes = [i32, i32, vp]
    L.ref_pass_compose.argtypes = [i32, i32, vp, vp, vp]
    L.ref_draw_jacobians.argtypes = [i32, i32, i32, vp, vp]
    L.ref_render_quads.argtypes = [vp, u32, vp, vp, vp, vp]
    L.ref_draw_indexmap.argtypes = [vp, u32, i32, i32, vp]
    L.ref_set_update_sources.argtypes = [vp]
    L.ref_set_update_sources.restype = None
    L.ref_draw_update.restype = u32
    L.ref_draw_update.argtypes = [vp, u32, i32, i32, vp, u32, vp]
    L.ref_draw_generate.restype = u32
    L.ref_draw_generate.argtypes = [i32, i32, vp, u32]
    L.ref_draw_copy.restype = u32
    L.ref_draw_copy.argtypes = [vp, u32, vp, u32, u32]
    L.ref_draw_extract.restype = u32
    L.ref_draw_extract.argtypes = [vp, u32, vp, u32]
    L.ref_se3_exp.argtypes = [vp, vp]
    L.ref_se3_log.argtypes = [vp, vp]
    L.ref_glsl_inverse.argtypes = [vp, vp]
    L.ref_glsl_pack.restype = f32
    L.ref_glsl_pack.argtypes = [f32, f32, f32]
    L.ref_glsl_slerp.argtypes = [vp, vp, f32, vp]
    L.ref_uses_libm.restype = i32
    _LIBS[variant] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_libm = C.CDLL("libm.so.6")
_libm.sinf.restype = C.c_float
_libm.sinf.argtypes = [C.c_float]


def _sinf(x):
    return np.float32(_libm.sinf(C.c_float(float(x))))


def rigid_inverse_f32(pose):
    """oracle/o_math.h om4_rigid_inverse: R^T, -R^T t in double, rounded once (row-major 4x4 in, row-major out).
    The reference uses Eigen's general `pose.inverse()` in float (SurfelMap.cpp:497,875); not compilable here."""
    m = np.asarray(pose, dtype=np.float32).astype(np.float64)
    R, t = m[:3, :3], m[:3, 3]
    out = np.zeros((4, 4), dtype=np.float32)
    out[:3, :3] = R.T.astype(np.float32)
    for r in range(3):
        s = (R[0, r] * t[0] + R[1, r] * t[1]) + R[2, r] * t[2]
        out[r, 3] = np.float32(-s)
    out[3, 3] = 1.0
    return out


class Ref:
    """The hot-path glow programs of the reference, with their uniforms set as the reference's host code sets them."""

    def __init__(self, params: SumaParams, variant: str = ""):
        self.L = lib(variant)
        self.p = params
        self._keep = {}
        p = params
        f = np.float32
        self.W, self.H = int(p.data_width), int(p.data_height)
        self.Wm, self.Hm = int(p.model_width), int(p.model_height)
        # SurfelMap::setParameters, SurfelMap.cpp:336-350
        vfov = f(abs(f(p.data_fov_up))) + f(abs(f(p.data_fov_down)))
        vpix = f(math.tan(0.5 * (float(vfov) * math.pi / 180.0) / self.H))
        hpix = f(math.tan(0.5 * (360.0 * math.pi / 180.0) / self.W))
        self.pixel_size = max(vpix, hpix)
        self.p_unstable = f(1.0) - f(p.p_stable)
        self.log_prior = f(math.log(float(f(p.p_prior)) / (1.0 - float(f(p.p_prior)))))
        self.log_unstable = f(math.log(float(self.p_unstable) / (1.0 - float(self.p_unstable))))

    # -- glUniform / glBindTexture helpers
    def _u(self, prog, name, value, kind="f", required=True):
        """kind: f / i scalar, m row-major 4x4 (sent column-major like Eigen::Matrix4f), v vector.  float32 values
        travel as doubles (exact) and are converted to the uniform's declared type."""
        if kind == "m":
            a = np.ascontiguousarray(np.asarray(value, dtype=np.float32).T).reshape(16).astype(np.float64)
        elif kind == "v":
            a = np.ascontiguousarray(np.asarray(value, dtype=np.float32)).astype(np.float64)
        elif kind == "i":
            a = np.array([int(value)], dtype=np.float64)
        else:
            a = np.array([np.float32(value)], dtype=np.float64)
        hits = self.L.ref_set_uniform(prog.encode(), name.encode(), _p(a))
        assert hits > 0 or not required, f"uniform {prog}.{name} not found"
        return hits

    def _tex(self, prog, name, arr, filt=NEAREST, required=True):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self._keep[(prog, name)] = arr
        h, w = arr.shape[0], arr.shape[1]
        ch = 1 if arr.ndim == 2 else arr.shape[2]
        hits = self.L.ref_bind_texture(prog.encode(), name.encode(), _p(arr), w, h, ch, filt)
        assert hits > 0 or not required, f"sampler {prog}.{name} not found"

    def _buf(self, prog, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self._keep[(prog, name)] = arr
        assert self.L.ref_bind_buffer(prog.encode(), name.encode(), _p(arr), arr.size // 4) > 0

    def _proj_uniforms(self, prog, model=False, wh=True):
        p = self.p
        up, down = (p.model_fov_up, p.model_fov_down) if model else (p.data_fov_up, p.data_fov_down)
        mn, mx = (p.model_min_depth, p.model_max_depth) if model else (p.min_depth, p.max_depth)
        self._u(prog, "fov_up", abs(np.float32(up)))
        self._u(prog, "fov_down", abs(np.float32(down)))
        self._u(prog, "min_depth", mn, required=False)
        self._u(prog, "max_depth", mx, required=False)
        if wh:
            self._u(prog, "width", self.Wm if model else self.W, required=False)
            self._u(prog, "height", self.Hm if model else self.H, required=False)

    # -- K1-K3  Preprocessing::process (Preprocessing.cpp:120-339); the filter passes (:150,160-166,191-236) run with
    #    the texture's own sampling state, which `filter_sampling` names (suma_types.h)
    def preprocess(self, points, labels, probs, timestamp):
        p = self.p
        W, H = self.W, self.H
        points = np.ascontiguousarray(points, dtype=np.float32)
        n = points.shape[0]
        # vertex fetch with the byte offsets of Preprocessing.cpp:142-145 (quirk B-1); reads past the end yield 0 here
        lab = np.zeros(n, dtype=np.float32)
        prb = np.zeros(n, dtype=np.float32)
        lo, po = int(p.label_offset), int(p.prob_offset)
        if labels is not None and n > lo:
            lab[: n - lo] = np.asarray(labels, dtype=np.float32)[lo:]
        if probs is not None and n > po:
            prb[: n - po] = np.asarray(probs, dtype=np.float32)[po:]
        # Preprocessing::setParameters, Preprocessing.cpp:92-100; isfirst :176-179
        self._proj_uniforms("gen_vertexmap")
        self._u("gen_vertexmap", "isfirst", 1 if timestamp < 10 else 0, "i")
        vmap = np.zeros((H, W, 4), dtype=np.float32)
        smap = np.zeros((H, W, 4), dtype=np.float32)
        own = NEAREST if int(p.filter_sampling) == 1 else (LINEAR | CLAMP_TO_EDGE)
        if p.avg_vertexmap:
            temp = np.zeros((H, W, 4), dtype=np.float32)
            self.L.ref_draw_vertexmap_blend(_p(points), _p(lab), _p(prb), n, W, H, _p(temp), _p(smap))
            self._tex("avg_vertexmap", "in_vertexmap", temp, own)
            self.L.ref_pass_avg_vertexmap(W, H, _p(vmap))
        else:
            self.L.ref_draw_vertexmap(_p(points), _p(lab), _p(prb), n, W, H, _p(vmap), _p(smap))
        if p.filter_vertexmap:
            prog = "bilateral_filter"
            self._u(prog, "width", W)
            self._u(prog, "height", H)
            self._u(prog, "sigma_space", p.bilateral_sigma_space)
            self._u(prog, "sigma_range", p.bilateral_sigma_range)
            self._tex(prog, "in_vertexmap", vmap, own)
            temp = np.zeros((H, W, 4), dtype=np.float32)
            self.L.ref_pass_bilateral(W, H, _p(temp))
            if p.use_filtered_vertexmap:
                vmap = temp  # Preprocessing.cpp:234
        nmap, _, refined = self.normals_and_labels(vmap, smap)
        return vmap, nmap, refined

    def normals_and_labels(self, vmap, smap):
        """passes 2 and 3 of Preprocessing::process on given maps: (normal map, eroded labels, refined labels)"""
        W, H = self.W, self.H
        vmap = np.ascontiguousarray(vmap, dtype=np.float32)
        smap = np.ascontiguousarray(smap, dtype=np.float32)
        # pass 2 (Preprocessing.cpp:238-279): sampler NEAREST + CLAMP_TO_BORDER (:68-70)
        self._tex("gen_normalmap", "vertex_map", vmap, NEAREST)
        self._tex("gen_normalmap", "semantic_map", smap, NEAREST)
        nmap = np.zeros((H, W, 4), dtype=np.float32)
        eroded = np.zeros((H, W, 4), dtype=np.float32)
        self.L.ref_pass_normalmap(W, H, _p(nmap), _p(eroded))
        # pass 3 (Preprocessing.cpp:281-327)
        self._tex("floodfill", "vertex_map", vmap, NEAREST)
        self._tex("floodfill", "semantic_map", eroded, NEAREST)
        refined = np.zeros((H, W, 4), dtype=np.float32)
        self.L.ref_pass_floodfill(W, H, _p(refined))
        return nmap, eroded, refined

    # -- K6  Frame2Model::jacobianProducts (Frame2Model.cpp:136-261; uniforms :65-110,194-195)
    def jacobians(self, cur, model, pose, iteration, entries_per_kernel=64, gates=None):
        p = self.p
        prog = "Frame2Model_jacobians"
        max_angle, max_dist = gates if gates else (p.icp_max_angle, p.icp_max_distance)
        self._u(prog, "angle_thresh", np.float32(math.cos(float(np.float32(max_angle)) * math.pi / 180.0)))
        self._u(prog, "distance_thresh", max_dist)
        self._u(prog, "weight_function", p.weight_function, "i")
        self._u(prog, "factor", p.factor)
        self._u(prog, "fov_up", abs(np.float32(p.data_fov_up)))
        self._u(prog, "fov_down", abs(np.float32(p.data_fov_down)))
        self._u(prog, "entries_per_kernel", entries_per_kernel, "i")
        self._u(prog, "iteration", iteration, "i")
        self._u(prog, "pose", np.asarray(pose, dtype=np.float64).astype(np.float32), "m")  # pose_.cast<float>()
        filt = LINEAR if p.bilinear_sampling else NEAREST  # Frame2Model.cpp:101-109, one sampler on all six units
        self._tex(prog, "vertex_model", model[0], filt)
        self._tex(prog, "normal_model", model[1], filt)
        self._tex(prog, "semantic_model", model[2], filt)
        self._tex(prog, "vertex_data", cur[0], filt)
        self._tex(prog, "normal_data", cur[1], filt)
        self._tex(prog, "semantic_data", cur[2], filt)
        blend = np.zeros(48, dtype=np.float32)
        fix = np.zeros(48, dtype=np.int64)
        self.L.ref_draw_jacobians(self.W, self.H, entries_per_kernel, _p(blend), _p(fix))
        return blend, fix

    # -- K4  vertex + geometry stage of SurfelMap::render* (SurfelMap.cpp:847-1165)
    def render_quads(self, surfels, poses, pose, conf_threshold, render_old, timestamp_threshold):
        prog = "render_surfels"
        self._proj_uniforms(prog, model=True, wh=False)
        self._u(prog, "use_stability", self.p.use_stability, "i")
        self._u(prog, "conf_threshold", conf_threshold)
        self._u(prog, "timestamp_threshold", timestamp_threshold, "i")
        self._u(prog, "render_old_surfels", 1 if render_old else 0, "i")
        self._u(prog, "inv_pose", rigid_inverse_f32(pose), "m")
        self._buf(prog, "poseBuffer", poses)
        surfels = np.ascontiguousarray(surfels, dtype=SURFEL_DTYPE)
        n = surfels.shape[0]
        emitted = np.zeros(n, dtype=np.uint8)
        pos = np.zeros((n, 4, 4), dtype=np.float32)
        tex = np.zeros((n, 4, 2), dtype=np.float32)
        attr = np.zeros((n, 13), dtype=np.float32)
        self.L.ref_render_quads(_p(surfels), n, _p(emitted), _p(pos), _p(tex), _p(attr))
        return emitted, pos, tex, attr

    # -- K5  render_compose.frag (SurfelMap.cpp:911-940)
    def compose(self, old, new):
        prog = "render_compose"
        self._u(prog, "max_distance", self.p.max_loop_closure_distance)
        for k, nm in enumerate(("vertexmap", "normalmap", "semanticmap")):
            self._tex(prog, "old_" + nm, old[k], NEAREST)
            self._tex(prog, "new_" + nm, new[k], NEAREST)
        out = [np.zeros((self.Hm, self.Wm, 4), dtype=np.float32) for _ in range(3)]
        self.L.ref_pass_compose(self.Wm, self.Hm, _p(out[0]), _p(out[1]), _p(out[2]))
        return out

    # -- K7  SurfelMap::renderIndexmap (SurfelMap.cpp:586-604; uniforms :144-150)
    def indexmap(self, surfels, poses, pose):
        prog = "gen_indexmap"
        self._proj_uniforms(prog)
        self._u(prog, "pose", pose, "m")
        self._u(prog, "inv_pose", rigid_inverse_f32(pose), "m")
        self._buf(prog, "poseBuffer", poses)
        surfels = np.ascontiguousarray(surfels, dtype=SURFEL_DTYPE)
        out = np.zeros((self.H, self.W), dtype=np.float32)
        self.L.ref_draw_indexmap(_p(surfels), surfels.shape[0], self.W, self.H, _p(out))
        return out

    # -- K8  SurfelMap::generateDataSurfels (SurfelMap.cpp:606-619; uniforms :378-397)
    def radius_conf(self, vmap, nmap):
        p = self.p
        prog = "init_radiusConf"
        self._proj_uniforms(prog, wh=False)
        self._u(prog, "pixel_size", self.pixel_size)
        self._u(prog, "confidence_mode", p.confidence_mode, "i")
        self._u(prog, "min_radius", p.min_radius)
        self._u(prog, "max_radius", p.max_radius)
        # :394-397 std::cos(radians(float(max_angle)))
        self._u(prog, "angle_thresh", np.float32(math.cos(float(np.float32(float(np.float32(p.max_angle)) * math.pi / 180.0)))))
        # the map's sampler (SurfelMap.cpp:168-170) is MIN NEAREST / MAG LINEAR; fetches are at texel centres
        self._tex(prog, "vertex_map", vmap, LINEAR)
        self._tex(prog, "normal_map", nmap, LINEAR)
        out = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self.L.ref_draw_radius_conf(self.W, self.H, _p(out))
        return out

    # -- K9  SurfelMap::updateSurfels, first draw (SurfelMap.cpp:621-644; uniforms :399-438)
    def update_uniforms(self, pose, timestamp):
        """the uniforms of update_program_ as SurfelMap.cpp:399-438 and :626-628 set them: [(name, value, kind)]"""
        p = self.p
        up, down = p.data_fov_up, p.data_fov_down
        return [
            ("fov_up", abs(np.float32(up)), "f"), ("fov_down", abs(np.float32(down)), "f"),
            ("min_depth", p.min_depth, "f"), ("max_depth", p.max_depth, "f"),
            ("width", self.W, "f"), ("height", self.H, "f"),
            ("pixel_size", self.pixel_size, "f"),
            ("distance_thresh", p.map_max_distance, "f"),
            ("angle_thresh", _sinf(np.float32(np.float32(math.pi) / np.float32(180.0)) * np.float32(p.map_max_angle)), "f"),
            ("confidence_mode", p.confidence_mode, "i"), ("unstable_age", p.unstable_age, "i"),
            ("p_stable", p.p_stable, "f"), ("p_unstable", self.p_unstable, "f"), ("p_prior", p.p_prior, "f"),
            ("log_prior", self.log_prior, "f"), ("log_unstable", self.log_unstable, "f"),
            ("sigma_angle", p.sigma_angle, "f"), ("sigma_distance", p.sigma_distance, "f"),
            ("confidence_threshold", p.confidence_threshold, "f"),
            ("min_radius", 0.0, "f"),  # SurfelMap.cpp:422: the update program keeps 0
            ("max_weight", p.max_weight, "f"),
            ("weighting_scheme", p.weighting_scheme, "i"), ("averaging_scheme", p.averaging_scheme, "i"),
            ("update_always", p.update_always, "i"), ("active_timestamps", p.active_timestamps, "i"),
            ("use_stability", p.use_stability, "i"),
            ("pose", pose, "m"), ("inv_pose", rigid_inverse_f32(pose), "m"), ("timestamp", timestamp, "i"),
        ]

    def update(self, surfels, poses, pose, timestamp, frame, radconf, index_map_float, sources=False):
        """sources=True: also returns, per output record, the index of the input surfel it came from"""
        p = self.p
        prog = "update_surfels"
        optional = ("min_depth", "max_depth", "width", "height")
        for name, value, kind in self.update_uniforms(pose, timestamp):
            self._u(prog, name, value, kind, required=name not in optional)
        self._buf(prog, "poseBuffer", poses)
        self._tex(prog, "vertex_map", frame[0], LINEAR)
        self._tex(prog, "normal_map", frame[1], LINEAR)
        self._tex(prog, "semantic_map_in", frame[2], LINEAR)
        self._tex(prog, "radiusConfidence_map", radconf, LINEAR)
        self._tex(prog, "index_map", index_map_float, LINEAR)
        surfels = np.ascontiguousarray(surfels, dtype=SURFEL_DTYPE)
        cap = int(p.max_surfels)
        out = np.zeros(max(surfels.shape[0], 1), dtype=SURFEL_DTYPE)
        mask = np.zeros((self.H, self.W, 4), dtype=np.float32)
        src = np.zeros(out.shape[0], dtype=np.uint32)
        if sources:
            self.L.ref_set_update_sources(_p(src))
        n = self.L.ref_draw_update(_p(surfels), surfels.shape[0], self.W, self.H, _p(out), min(cap, out.shape[0]), _p(mask))
        if sources:
            return out[:n].copy(), mask, src[:n].copy()
        return out[:n].copy(), mask

    # -- K10  SurfelMap::updateSurfels, second draw (SurfelMap.cpp:646-664; uniforms :360-376)
    def generate_uniforms(self, pose, timestamp):
        """the uniforms of initialize_program_ as SurfelMap.cpp:360-376 and :650-652 set them: [(name, value, kind)]"""
        p = self.p
        return [("fov_up", abs(np.float32(p.data_fov_up)), "f"), ("fov_down", abs(np.float32(p.data_fov_down)), "f"),
                ("min_depth", p.min_depth, "f"), ("max_depth", p.max_depth, "f"), ("width", self.W, "f"), ("height", self.H, "f"),
                ("pixel_size", self.pixel_size, "f"), ("log_prior", self.log_prior, "f"), ("pose", pose, "m"),
                ("inv_pose", rigid_inverse_f32(pose), "m"), ("timestamp", timestamp, "i")]

    def generate(self, frame, radconf, integrated4, pose, timestamp):
        prog = "gen_surfels"
        for name, value, kind in self.generate_uniforms(pose, timestamp):
            self._u(prog, name, value, kind, required=name not in ("min_depth", "max_depth", "width", "height", "inv_pose"))
        self._tex(prog, "vertex_map", frame[0], LINEAR)
        self._tex(prog, "normal_map", frame[1], LINEAR)
        self._tex(prog, "semantic_map", frame[2], LINEAR)
        self._tex(prog, "radiusConfidence_map", radconf, LINEAR)
        self._tex(prog, "measurementIntegrated_map", integrated4, LINEAR)
        # model_semantic_map: unit 9 is never bound (SurfelMap.cpp:370, quirk B-8) -> reads the border / zero
        cap = 2 * self.W * self.H  # SurfelMap.cpp:57
        out = np.zeros(cap, dtype=SURFEL_DTYPE)
        n = self.L.ref_draw_generate(self.W, self.H, _p(out), cap)
        return out[:n].copy()

    # -- K11  SurfelMap::copySurfels (SurfelMap.cpp:667-698)
    def copy(self, updated, data, poses, center, extent):
        prog = "copy_surfels"
        self._u(prog, "submap_center", np.asarray(center, dtype=np.float32), "v")
        self._u(prog, "submap_extent", extent)
        self._buf(prog, "poseBuffer", poses)
        cap = int(self.p.max_surfels)
        out = np.zeros(max(1, min(cap, updated.shape[0] + data.shape[0])), dtype=SURFEL_DTYPE)
        n = 0
        for src in (updated, data):
            src = np.ascontiguousarray(src, dtype=SURFEL_DTYPE)
            n = self.L.ref_draw_copy(_p(src), src.shape[0], _p(out), n, out.shape[0])
        return out[:n].copy()

    # -- K12  SurfelMap::extractSurfels (SurfelMap.cpp:708-742)
    def extract(self, surfels, poses, center, extent, cap=500000):
        prog = "extract_surfels"
        self._u(prog, "submap_center", np.asarray(center, dtype=np.float32), "v")
        self._u(prog, "submap_extent", extent)
        self._buf(prog, "poseBuffer", poses)
        surfels = np.ascontiguousarray(surfels, dtype=SURFEL_DTYPE)
        out = np.zeros(max(1, min(cap, surfels.shape[0])), dtype=SURFEL_DTYPE)
        n = self.L.ref_draw_extract(_p(surfels), surfels.shape[0], _p(out), out.shape[0])
        return out[:n].copy()


def se3_exp(x, variant=""):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros((4, 4), dtype=np.float64)
    lib(variant).ref_se3_exp(_p(x), _p(T))
    return T.T.copy()


def se3_log(T, variant=""):
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T)
    x = np.zeros(6, dtype=np.float64)
    lib(variant).ref_se3_log(_p(Tc), _p(x))
    return x


def glsl_inverse(m, variant=""):
    mc = np.ascontiguousarray(np.asarray(m, dtype=np.float32).T)
    out = np.zeros((4, 4), dtype=np.float32)
    lib(variant).ref_glsl_inverse(_p(mc), _p(out))
    return out.T.copy()


def glsl_slerp(v0, v1, w, variant=""):
    a = np.ascontiguousarray(v0, dtype=np.float32)
    b = np.ascontiguousarray(v1, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib(variant).ref_glsl_slerp(_p(a), _p(b), C.c_float(w), _p(out))
    return out
