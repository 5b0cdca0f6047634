"""TEST INFRASTRUCTURE -- ctypes loader for the CPU oracle (oracle/libsuma_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.  The oracle is a CPU restatement of the reference's GLSL path
(pinned to the reference's own shader sources compiled with g++: oracle/_ref, tests/test_ref_shaders.py; the
reference ships no golden vectors of its own).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from semantic_suma_amd.types import ACC_WORDS, SURFEL_DTYPE, IcpStats, SumaParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

class OraLoopResult(C.Structure):
    """ora_loop_result (oracle/suma_oracle.h)"""
    _fields_ = [("gn_pose", C.c_double * 16), ("after_minimize", IcpStats), ("passed", C.c_int32),
                ("pose_old", C.c_float * 16), ("composed", IcpStats), ("JtJ", C.c_double * 36)]


class OraLoopTrack(C.Structure):
    """ora_loop_track (oracle/suma_oracle.h)"""
    _fields_ = [("increment_old", C.c_double * 16), ("after_minimize", IcpStats), ("increment_difference", C.c_float),
                ("passed", C.c_int32), ("pose_old", C.c_double * 16), ("composed", IcpStats), ("JtJ", C.c_double * 36)]


c_f4p = C.POINTER(C.c_float)
c_f8p = C.POINTER(C.c_double)


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (seconds)."""
    tgt = os.path.join(_HERE, "libsuma_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("suma_detmath.h", "suma_types.h")]
    if force or not os.path.exists(tgt) or any(os.path.getmtime(s) > os.path.getmtime(tgt) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def build_native() -> bool:
    """-O3 -march=native build for bench.py's cpu_baseline leg (must be compiled on the host it runs on).
    Returns False when no compiler is available there (the -O2 library is used instead)."""
    try:
        subprocess.check_call(["make", "-C", _HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return os.path.exists(os.path.join(_HERE, "libsuma_oracle_native.so"))
    except (OSError, subprocess.CalledProcessError):
        return False


def lib(variant: str = ""):
    """variant '' = deterministic-math oracle (-O2), 'native' = same sources -O3 -march=native, 'libm' = glibc
    transcendental functions."""
    if variant in _LIBS:
        return _LIBS[variant]
    name = "libsuma_oracle.so" if not variant else f"libsuma_oracle_{variant}.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ora_create.restype = vp
    L.ora_create.argtypes = [C.POINTER(SumaParams)]
    L.ora_destroy.argtypes = [vp]
    L.ora_set_params.argtypes = [vp, C.POINTER(SumaParams)]
    L.ora_set_threads.argtypes = [vp, C.c_int]
    L.ora_frame_create.restype = vp
    L.ora_frame_create.argtypes = [C.c_uint32, C.c_uint32]
    L.ora_frame_destroy.argtypes = [vp]
    L.ora_frame_map.restype = vp
    L.ora_frame_map.argtypes = [vp, C.c_int]
    L.ora_frame_copy.argtypes = [vp, vp]
    L.ora_preprocess.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.ora_icp_jacobian_products.restype = C.c_double
    L.ora_icp_jacobian_products.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, vp, vp, C.POINTER(IcpStats)]
    L.ora_icp_minimize.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(IcpStats)]
    L.ora_icp_minimize_from.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(IcpStats)]
    L.ora_map_reset.argtypes = [vp]
    L.ora_map_update.argtypes = [vp, vp, vp]
    L.ora_map_render.argtypes = [vp, vp, vp, C.c_float, vp]
    L.ora_map_render_active.argtypes = [vp, vp, C.c_float]
    L.ora_map_render_inactive.argtypes = [vp, vp, C.c_float]
    L.ora_map_render_composed.argtypes = [vp, vp, vp, C.c_float]
    L.ora_map_frame.restype = vp
    L.ora_map_frame.argtypes = [vp, C.c_int]
    L.ora_map_update_poses.argtypes = [vp, vp, C.c_uint32]
    L.ora_map_size.restype = C.c_uint32
    L.ora_map_size.argtypes = [vp]
    L.ora_map_timestamp.restype = C.c_uint32
    L.ora_map_timestamp.argtypes = [vp]
    L.ora_map_surfels.restype = vp
    L.ora_map_surfels.argtypes = [vp]
    L.ora_map_upload.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    for n in ("ora_map_index_map", "ora_map_radius_conf", "ora_map_integrated"):
        getattr(L, n).restype = vp
        getattr(L, n).argtypes = [vp]
    for n in ("ora_map_last_updated_count", "ora_map_last_new_count", "ora_map_cached_surfels"):
        getattr(L, n).restype = C.c_uint32
        getattr(L, n).argtypes = [vp]
    L.ora_map_submap_origin.argtypes = [vp, vp]
    for n in ("ora_map_updated_surfels", "ora_map_data_surfels", "ora_map_poses"):
        getattr(L, n).restype = vp
        getattr(L, n).argtypes = [vp]
    L.ora_map_pending_extractions.restype = C.c_uint32
    L.ora_map_pending_extractions.argtypes = [vp]
    L.ora_map_cache_tile.restype = vp
    L.ora_map_cache_tile.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]
    L.ora_map_last_extraction.restype = C.c_uint32
    L.ora_map_last_extraction.argtypes = [vp, vp]
    L.ora_debug_render_quads.argtypes = [vp, vp, C.c_float, C.c_int, C.c_int32, vp, vp, vp]
    L.ora_debug_raster_quads.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_uint32, C.c_int, C.c_int, vp]
    L.ora_debug_depth24.argtypes = [vp, C.c_uint32, vp]
    L.ora_pipeline_create.restype = vp
    L.ora_pipeline_create.argtypes = [C.POINTER(SumaParams)]
    L.ora_pipeline_destroy.argtypes = [vp]
    L.ora_pipeline_ctx.restype = vp
    L.ora_pipeline_ctx.argtypes = [vp]
    L.ora_pipeline_process_scan.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_int32]
    L.ora_pipeline_pose.argtypes = [vp, vp]
    L.ora_pipeline_last_increment.argtypes = [vp, vp]
    L.ora_pipeline_last_stats.argtypes = [vp, C.POINTER(IcpStats)]
    L.ora_pipeline_track_loss.restype = C.c_uint32
    L.ora_pipeline_track_loss.argtypes = [vp]
    L.ora_pipeline_frame.restype = vp
    L.ora_pipeline_frame.argtypes = [vp, C.c_int]
    L.ora_se3_exp.argtypes = [vp, vp]
    L.ora_se3_log.argtypes = [vp, vp]
    L.ora_pipeline_begin_scan.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.ora_pipeline_update_pose.argtypes = [vp, C.c_int32]
    L.ora_pipeline_update_map.argtypes = [vp]
    L.ora_pipeline_integrate_loop_closures.argtypes = [vp, vp, C.c_uint32, vp]
    L.ora_pipeline_set_pose_old.argtypes = [vp, vp]
    L.ora_pipeline_get_pose.argtypes = [vp, C.c_int, vp]
    L.ora_pipeline_verify_loop_closure.argtypes = [vp, vp, vp, C.c_uint32, C.c_float, C.c_float, C.POINTER(OraLoopResult)]
    L.ora_pipeline_track_loop_closure.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.POINTER(OraLoopTrack)]
    L.ora_loop_closure_verify.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, C.c_float, C.c_float, C.c_float,
                                          C.POINTER(OraLoopResult)]
    L.ora_loop_closure_track.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_double, C.c_double, C.c_double,
                                         C.POINTER(OraLoopTrack)]
    L.ora_solve6.argtypes = [vp, vp, vp]
    _LIBS[variant] = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OracleFrame:
    """Host-side Frame: three H x W x 4 float32 maps (row 0 = bottom row)."""

    def __init__(self, L, width, height, handle=None):
        self.L, self.width, self.height = L, width, height
        self.owned = handle is None
        self.h = L.ora_frame_create(width, height) if handle is None else handle

    def map(self, which) -> np.ndarray:
        return _view(self.L.ora_frame_map(self.h, which), (self.height, self.width, 4), np.float32)

    @property
    def vertex(self):
        return self.map(0)

    @property
    def normal(self):
        return self.map(1)

    @property
    def semantic(self):
        return self.map(2)

    def set(self, vertex, normal, semantic):
        self.vertex[...] = vertex
        self.normal[...] = normal
        self.semantic[...] = semantic

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.L.ora_frame_destroy(self.h)
            self.h = None


class Oracle:
    """One oracle context = Preprocessing + Frame2Model + LieGaussNewton + SurfelMap state."""

    def __init__(self, params: SumaParams, variant: str = "", threads: int = 1, handle=None, keepalive=None):
        self.L = lib(variant)
        self.params = params
        self.owned = handle is None
        self.h = self.L.ora_create(C.byref(params)) if handle is None else handle
        self._keepalive = keepalive
        if threads != 1:
            self.L.ora_set_threads(self.h, threads)

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.L.ora_destroy(self.h)
            self.h = None

    def set_params(self, params):
        self.params = params
        self.L.ora_set_params(self.h, C.byref(params))

    def set_threads(self, n):
        self.L.ora_set_threads(self.h, n)

    def frame(self, model=False) -> OracleFrame:
        p = self.params
        return OracleFrame(self.L, p.model_width if model else p.data_width, p.model_height if model else p.data_height)

    # -- Preprocessing::process
    def preprocess(self, points, labels, probs, timestamp, out: OracleFrame):
        points = np.ascontiguousarray(points, dtype=np.float32)
        labels = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        probs = None if probs is None else np.ascontiguousarray(probs, dtype=np.float32)
        self.L.ora_preprocess(self.h, _ptr(points), _ptr(labels), _ptr(probs), points.shape[0], timestamp, out.h)
        return out

    # -- Frame2Model::jacobianProducts
    def jacobian_products(self, current, model, pose, iteration=0):
        pose = np.ascontiguousarray(np.asarray(pose, dtype=np.float64).T)  # column-major
        acc = np.zeros(ACC_WORDS, dtype=np.int64)
        JtJ = np.zeros((6, 6), dtype=np.float64)
        Jtr = np.zeros(6, dtype=np.float64)
        st = IcpStats()
        F = self.L.ora_icp_jacobian_products(self.h, current.h, model.h, _ptr(pose), iteration, _ptr(acc), _ptr(JtJ),
                                             _ptr(Jtr), C.byref(st))
        return F, acc, JtJ, Jtr, st

    # -- LieGaussNewton::minimize
    def minimize(self, current, model, T0, history_cap=64, iteration0=0):
        """iteration0: Frame2Model::iteration_ at the start (0 right behind a setData, Frame2Model.cpp:117-123)"""
        T0 = np.ascontiguousarray(np.asarray(T0, dtype=np.float64).T)
        T = np.zeros((4, 4), dtype=np.float64)
        hist = np.zeros((history_cap, 4, 4), dtype=np.float64)
        nh = C.c_uint32(0)
        st = IcpStats()
        self.L.ora_icp_minimize_from(self.h, current.h, model.h, _ptr(T0), iteration0, _ptr(T), _ptr(hist), history_cap,
                                     C.byref(nh), C.byref(st))
        n = min(nh.value, history_cap)
        return T.T.copy(), hist[:n].transpose(0, 2, 1).copy(), st

    # -- SurfelMap
    def map_reset(self):
        self.L.ora_map_reset(self.h)

    def map_update(self, pose, frame):
        pose = np.ascontiguousarray(np.asarray(pose, dtype=np.float32).T)
        self.L.ora_map_update(self.h, _ptr(pose), frame.h)

    def map_render(self, pose_old, pose_new, conf_threshold, out):
        po = np.ascontiguousarray(np.asarray(pose_old, dtype=np.float32).T)
        pn = np.ascontiguousarray(np.asarray(pose_new, dtype=np.float32).T)
        self.L.ora_map_render(self.h, _ptr(po), _ptr(pn), conf_threshold, out.h)
        return out

    def map_render_active(self, pose, conf_threshold):
        p = np.ascontiguousarray(np.asarray(pose, dtype=np.float32).T)
        self.L.ora_map_render_active(self.h, _ptr(p), conf_threshold)

    def map_render_inactive(self, pose, conf_threshold):
        p = np.ascontiguousarray(np.asarray(pose, dtype=np.float32).T)
        self.L.ora_map_render_inactive(self.h, _ptr(p), conf_threshold)

    def map_render_composed(self, pose_old, pose_new, conf_threshold):
        po = np.ascontiguousarray(np.asarray(pose_old, dtype=np.float32).T)
        pn = np.ascontiguousarray(np.asarray(pose_new, dtype=np.float32).T)
        self.L.ora_map_render_composed(self.h, _ptr(po), _ptr(pn), conf_threshold)

    def map_frame(self, which) -> OracleFrame:
        p = self.params
        return OracleFrame(self.L, p.model_width, p.model_height, handle=self.L.ora_map_frame(self.h, which))

    def map_update_poses(self, poses):
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).transpose(0, 2, 1))
        self.L.ora_map_update_poses(self.h, _ptr(poses), poses.shape[0])

    def map_size(self):
        return self.L.ora_map_size(self.h)

    def map_timestamp(self):
        return self.L.ora_map_timestamp(self.h)

    def map_surfels(self) -> np.ndarray:
        n = self.map_size()
        if n == 0:
            return np.zeros(0, dtype=SURFEL_DTYPE)
        return _view(self.L.ora_map_surfels(self.h), (n,), SURFEL_DTYPE).copy()

    def map_upload(self, surfels, timestamp):
        surfels = np.ascontiguousarray(surfels, dtype=SURFEL_DTYPE)
        self.L.ora_map_upload(self.h, _ptr(surfels), surfels.shape[0], timestamp)

    def map_index_map(self):
        p = self.params
        return _view(self.L.ora_map_index_map(self.h), (p.data_height, p.data_width), np.uint32).copy()

    def map_radius_conf(self):
        p = self.params
        return _view(self.L.ora_map_radius_conf(self.h), (p.data_height, p.data_width, 4), np.float32).copy()

    def map_integrated(self):
        p = self.params
        return _view(self.L.ora_map_integrated(self.h), (p.data_height, p.data_width), np.uint8).copy()

    def map_counts(self):
        return self.L.ora_map_last_updated_count(self.h), self.L.ora_map_last_new_count(self.h)

    def map_cached_surfels(self):
        return self.L.ora_map_cached_surfels(self.h)

    # -- stage exports (tests/test_ref_shaders.py)
    def map_updated_surfels(self) -> np.ndarray:
        """K9 output of the last update (S' records, before K11)"""
        n = self.L.ora_map_last_updated_count(self.h)
        return _view(self.L.ora_map_updated_surfels(self.h), (max(n, 1),), SURFEL_DTYPE)[:n].copy()

    def map_data_surfels(self) -> np.ndarray:
        """K10 output of the last update (D records, before K11)"""
        n = self.L.ora_map_last_new_count(self.h)
        return _view(self.L.ora_map_data_surfels(self.h), (max(n, 1),), SURFEL_DTYPE)[:n].copy()

    def map_poses(self, n=None) -> np.ndarray:
        """pose table as stored (column-major 4x4 floats), n x 16"""
        n = self.params.max_poses if n is None else n
        return _view(self.L.ora_map_poses(self.h), (n, 16), np.float32).copy()

    def map_pending_extractions(self):
        return self.L.ora_map_pending_extractions(self.h)

    def map_cache_tile(self, i, j) -> np.ndarray:
        n = C.c_uint32(0)
        ptr = self.L.ora_map_cache_tile(self.h, i, j, C.byref(n))
        if not ptr or n.value == 0:
            return np.zeros(0, dtype=SURFEL_DTYPE)
        return _view(ptr, (n.value,), SURFEL_DTYPE).copy()

    def map_last_extraction(self):
        """(number of K12 extractions so far, (i, j) of the last one)"""
        ij = np.zeros(2, dtype=np.int32)
        n = self.L.ora_map_last_extraction(self.h, _ptr(ij))
        return n, (int(ij[0]), int(ij[1]))

    def debug_render_quads(self, pose, conf_threshold, mode, thr):
        """K4 vertex + geometry stage per surfel: (emitted, corners[n,4,3] in [0,1]^3, sensor-frame pos/normal[n,6])"""
        n = self.map_size()
        po = np.ascontiguousarray(np.asarray(pose, dtype=np.float32).T)
        emitted = np.zeros(max(n, 1), dtype=np.uint8)
        corners = np.zeros((max(n, 1), 4, 3), dtype=np.float32)
        pn = np.zeros((max(n, 1), 6), dtype=np.float32)
        self.L.ora_debug_render_quads(self.h, _ptr(po), conf_threshold, mode, thr, _ptr(emitted), _ptr(corners), _ptr(pn))
        return emitted[:n], corners[:n], pn[:n]

    def debug_raster_quads(self, W, H, corners, ids, use_disc=True, use_depth=True):
        """the triangle rasteriser alone on given quads ([n, 4, 3] corners in [0,1]^3): winner id per pixel, -1 = none"""
        corners = np.ascontiguousarray(corners, dtype=np.float32)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        win = np.zeros((H, W), dtype=np.int64)
        self.L.ora_debug_raster_quads(W, H, _ptr(corners), _ptr(ids), ids.shape[0], int(use_disc), int(use_depth), _ptr(win))
        return win

    def debug_depth24(self, zw):
        zw = np.ascontiguousarray(zw, dtype=np.float32)
        out = np.zeros(zw.shape[0], dtype=np.uint32)
        self.L.ora_debug_depth24(_ptr(zw), zw.shape[0], _ptr(out))
        return out

    def map_submap_origin(self):
        ij = np.zeros(2, dtype=np.int32)
        self.L.ora_map_submap_origin(self.h, _ptr(ij))
        return int(ij[0]), int(ij[1])


def _cmT(T, dtype):
    return np.ascontiguousarray(np.asarray(T, dtype=dtype).reshape(4, 4).T)


def _loop_results(res, n):
    out = []
    for k in range(n):
        r = res[k]
        out.append(dict(gn_pose=np.array(r.gn_pose[:]).reshape(4, 4).T.copy(), after_minimize=r.after_minimize.as_dict(),
                        passed=bool(r.passed), pose_old=np.array(r.pose_old[:], dtype=np.float32).reshape(4, 4).T.copy(),
                        composed=r.composed.as_dict(), JtJ=np.array(r.JtJ[:]).reshape(6, 6).T.copy()))
    return out


def _loop_track(r):
    return dict(increment_old=np.array(r.increment_old[:]).reshape(4, 4).T.copy(), after_minimize=r.after_minimize.as_dict(),
                increment_difference=float(r.increment_difference), passed=bool(r.passed),
                pose_old=np.array(r.pose_old[:]).reshape(4, 4).T.copy(), composed=r.composed.as_dict(),
                JtJ=np.array(r.JtJ[:]).reshape(6, 6).T.copy())


class OraclePipeline:
    """SurfelMapping::processScan: one call, or its phases with the loop-closure hooks between them."""

    def __init__(self, params: SumaParams, variant: str = "", threads: int = 1):
        self.L = lib(variant)
        self.params = params
        self.h = self.L.ora_pipeline_create(C.byref(params))
        self.ctx = Oracle(params, variant, handle=self.L.ora_pipeline_ctx(self.h), keepalive=self)
        if threads != 1:
            self.ctx.set_threads(threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_pipeline_destroy(self.h)
            self.h = None

    def process_scan(self, points, labels, probs, fixed_iterations=0):
        points = np.ascontiguousarray(points, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.float32)
        probs = np.ascontiguousarray(probs, dtype=np.float32)
        self.L.ora_pipeline_process_scan(self.h, _ptr(points), _ptr(labels), _ptr(probs), points.shape[0],
                                         fixed_iterations)

    def begin_scan(self, points, labels, probs):
        points = np.ascontiguousarray(points, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.float32)
        probs = np.ascontiguousarray(probs, dtype=np.float32)
        self.L.ora_pipeline_begin_scan(self.h, _ptr(points), _ptr(labels), _ptr(probs), points.shape[0])

    def update_pose(self, fixed_iterations=0):
        self.L.ora_pipeline_update_pose(self.h, fixed_iterations)

    def update_map(self):
        self.L.ora_pipeline_update_map(self.h)

    def integrate_loop_closures(self, poses, difference):
        P = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))
        D = _cmT(difference, np.float64)
        self.L.ora_pipeline_integrate_loop_closures(self.h, _ptr(P), P.shape[0], _ptr(D))

    def set_pose_old(self, pose_old):
        T = _cmT(pose_old, np.float64)
        self.L.ora_pipeline_set_pose_old(self.h, _ptr(T))

    def get_pose(self, which):
        T = np.zeros((4, 4), dtype=np.float64)
        self.L.ora_pipeline_get_pose(self.h, which, _ptr(T))
        return T.T.copy()

    def verify_loop_closure(self, pose_prior, initializations, min_valid_ratio=0.2, max_outlier_ratio=0.85):
        n = len(initializations)
        res = (OraLoopResult * n)()
        prior = _cmT(pose_prior, np.float64)
        inits = np.ascontiguousarray(np.stack([_cmT(T, np.float64) for T in initializations]))
        self.L.ora_pipeline_verify_loop_closure(self.h, _ptr(prior), _ptr(inits), n, min_valid_ratio, max_outlier_ratio, res)
        return _loop_results(res, n)

    def track_loop_closure(self, min_valid_ratio=0.2, max_outlier_ratio=0.85, max_increment_difference=0.1):
        r = OraLoopTrack()
        self.L.ora_pipeline_track_loop_closure(self.h, min_valid_ratio, max_outlier_ratio, max_increment_difference,
                                               C.byref(r))
        return _loop_track(r)

    def pose(self):
        T = np.zeros((4, 4), dtype=np.float64)
        self.L.ora_pipeline_pose(self.h, _ptr(T))
        return T.T.copy()

    def last_increment(self):
        T = np.zeros((4, 4), dtype=np.float64)
        self.L.ora_pipeline_last_increment(self.h, _ptr(T))
        return T.T.copy()

    def last_stats(self):
        st = IcpStats()
        self.L.ora_pipeline_last_stats(self.h, C.byref(st))
        return st

    def track_loss(self) -> int:
        """scans on which the frame-to-frame fallback ran (trackLoss_, SurfelMapping.cpp:441)"""
        return int(self.L.ora_pipeline_track_loss(self.h))

    def frame(self, which) -> OracleFrame:
        p = self.params
        w, h = (p.data_width, p.data_height) if which == 0 else (p.model_width, p.model_height)
        return OracleFrame(self.L, w, h, handle=self.L.ora_pipeline_frame(self.h, which))


def se3_exp(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros((4, 4), dtype=np.float64)
    lib().ora_se3_exp(_ptr(x), _ptr(T))
    return T.T.copy()


def se3_log(T):
    Tc = _cmT(T, np.float64)
    x = np.zeros(6, dtype=np.float64)
    lib().ora_se3_log(_ptr(Tc), _ptr(x))
    return x


def solve6(JtJ, Jtr):
    JtJ = np.ascontiguousarray(np.asarray(JtJ, dtype=np.float64).T)
    Jtr = np.ascontiguousarray(Jtr, dtype=np.float64)
    x = np.zeros(6, dtype=np.float64)
    lib().ora_solve6(_ptr(JtJ), _ptr(Jtr), _ptr(x))
    return x
