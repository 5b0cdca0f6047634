"""Host-side mirror of the reference's hot-path classes over the C-ABI (include/suma_hip.h).

The classes keep the names, argument meaning and error behaviour of the reference interfaces they
stand for (PRBonn/semantic_suma, src/core):

    Frame            src/core/Frame.h:21-79
    Preprocessing    src/core/Preprocessing.h:47-58          process(points, frame, labels, probs, timestamp)
    Frame2Model      src/core/Frame2Model.h:28-73 / Objective.h:14-82
    LieGaussNewton   src/core/LieGaussNewton.h:25-76         minimize(objective, T0), pose(), history()
    SurfelMap        src/core/SurfelMap.h:36-78              update / render* / *MapFrame / updatePoses / size
    SurfelMapping    src/core/SurfelMapping.h:47             processScan(scan)

Everything here is plumbing: numpy arrays in, ctypes calls into ``libsuma_hip.so`` (hand-written
gfx950 kernels), numpy arrays out.  There is no CPU fallback -- if the library is missing or no
MI355X is visible the constructors raise (the reference throws ``std::runtime_error`` in the
same situations, e.g. Frame2Model.cpp:132).  Matrices are exchanged as row-major numpy 4x4 and
converted to the column-major layout of Eigen at the boundary.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .types import ACC_WORDS, SURFEL_DTYPE, IcpStats, SumaParams

_HERE = os.path.dirname(os.path.abspath(__file__))
# SUMA_HIP_LIB selects another build of the same library (A/B timing of kernel variants in one GPU session)
LIB_PATH = os.environ.get("SUMA_HIP_LIB") or os.path.join(_HERE, "libsuma_hip.so")
_LIB = None


class SumaError(RuntimeError):
    """Raised for every negative return code of the C-ABI (the reference throws std::runtime_error)."""


class LoopResult(C.Structure):
    """suma_loop_result (include/suma_hip.h)"""
    _fields_ = [("gn_pose", C.c_double * 16), ("after_minimize", IcpStats), ("passed", C.c_int32),
                ("pose_old", C.c_float * 16), ("composed", IcpStats), ("JtJ", C.c_double * 36)]


class LoopTrack(C.Structure):
    """suma_loop_track (include/suma_hip.h)"""
    _fields_ = [("increment_old", C.c_double * 16), ("after_minimize", IcpStats), ("increment_difference", C.c_float),
                ("passed", C.c_int32), ("pose_old", C.c_double * 16), ("composed", IcpStats), ("JtJ", C.c_double * 36)]


class ScanRef(C.Structure):
    """suma_scan_ref (include/suma_runner.h)"""
    _fields_ = [("points", C.c_void_p), ("labels", C.c_void_p), ("probs", C.c_void_p), ("n", C.c_uint32)]


class SequenceJob(C.Structure):
    """suma_sequence_job"""
    _fields_ = [("scans", C.POINTER(ScanRef)), ("n_scans", C.c_uint32), ("on_device", C.c_int32)]


class SequenceResult(C.Structure):
    """suma_sequence_result"""
    _fields_ = [("status", C.c_int32), ("scans_done", C.c_uint32), ("map_surfels", C.c_uint32), ("track_loss", C.c_uint32),
                ("end_pose", C.c_double * 16), ("seconds", C.c_double), ("error", C.c_char * 160)]


class HypothesisJob(C.Structure):
    """suma_hypothesis_job"""
    _fields_ = [("scans", C.POINTER(ScanRef)), ("n_scans", C.c_uint32), ("on_device", C.c_int32),
                ("perturbations", C.c_void_p), ("n_hyp", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint32)


class IcpObjective(C.Structure):
    """suma_icp_objective (include/suma_hip.h): the parameters one Frame2Model object owns"""
    _fields_ = [("icp_max_distance", C.c_float), ("icp_max_angle", C.c_float), ("weight_function", C.c_int32),
                ("factor", C.c_float), ("bilinear_sampling", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("bytes", C.c_double)]


def lib():
    """Load libsuma_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C semantic_suma_amd/csrc``)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SumaError(f"{LIB_PATH} not found: build it with `make -C semantic_suma_amd/csrc` "
                        "(there is no CPU fallback for the HIP path)")
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
    pp = C.POINTER(vp)
    L.suma_version.restype = C.c_char_p
    L.suma_last_error.restype = C.c_char_p
    L.suma_last_error.argtypes = [vp]
    L.suma_ctx_create.argtypes = [C.POINTER(SumaParams), C.c_int, pp]
    L.suma_ctx_destroy.argtypes = [vp]
    L.suma_ctx_destroy.restype = None
    L.suma_set_params.argtypes = [vp, C.POINTER(SumaParams)]
    L.suma_synchronize.argtypes = [vp]
    L.suma_ctx_stream.restype = vp
    L.suma_ctx_stream.argtypes = [vp]
    L.suma_frame_create.argtypes = [vp, u32, u32, pp]
    L.suma_frame_destroy.argtypes = [vp]
    L.suma_frame_destroy.restype = None
    L.suma_frame_copy.argtypes = [vp, vp, vp]
    L.suma_frame_download.argtypes = [vp, vp, C.c_int, vp]
    L.suma_frame_upload.argtypes = [vp, vp, C.c_int, vp]
    L.suma_frame_width.restype = u32
    L.suma_frame_width.argtypes = [vp]
    L.suma_frame_height.restype = u32
    L.suma_frame_height.argtypes = [vp]
    L.suma_frame_device_ptr.restype = vp
    L.suma_frame_device_ptr.argtypes = [vp, C.c_int]
    L.suma_frame_swap.argtypes = [vp, vp, vp]
    L.suma_frame_export.argtypes = [vp, vp, C.c_int, pp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.suma_icp_set_objective.argtypes = [vp, C.POINTER(IcpObjective)]
    L.suma_icp_information.argtypes = [vp, vp]
    L.suma_map_export_surfels.argtypes = [vp, pp, C.POINTER(u32)]
    L.suma_map_export_data_surfels.argtypes = [vp, pp, C.POINTER(u32), C.POINTER(u32)]
    L.suma_pipeline_prefetch_scan.argtypes = [vp, vp, vp, vp, u32]
    L.suma_pipeline_process_prefetched.argtypes = [vp, i32]
    L.suma_pipeline_process_scan_async.argtypes = [vp, vp, vp, vp, u32, i32]
    L.suma_device_download.argtypes = [vp, vp, vp, C.c_uint64]
    L.suma_preprocess.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.suma_preprocess_device.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.suma_icp_set_data.argtypes = [vp, vp, vp]
    L.suma_icp_jacobian_products.argtypes = [vp, vp, u32, vp, vp, vp, C.POINTER(IcpStats)]
    L.suma_icp_minimize.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(IcpStats)]
    L.suma_icp_history.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.suma_icp_history_sequence.argtypes = [vp]
    L.suma_icp_history_sequence.restype = C.c_uint64
    L.suma_frame_touch.argtypes = [vp, vp]
    L.suma_pipeline_minimize_stats.argtypes = [vp, C.POINTER(IcpStats)]
    L.suma_icp_minimize_batch.argtypes = [vp, vp, u32, vp, vp]
    L.suma_map_reset.argtypes = [vp]
    L.suma_map_update.argtypes = [vp, vp, vp]
    L.suma_map_render.argtypes = [vp, vp, vp, f32, vp]
    L.suma_map_render_active.argtypes = [vp, vp, f32]
    L.suma_map_render_inactive.argtypes = [vp, vp, f32]
    L.suma_map_render_composed.argtypes = [vp, vp, vp, f32]
    L.suma_map_frame.restype = vp
    L.suma_map_frame.argtypes = [vp, C.c_int]
    L.suma_map_update_poses.argtypes = [vp, vp, u32]
    L.suma_map_size.argtypes = [vp, C.POINTER(u32)]
    L.suma_map_timestamp.argtypes = [vp, C.POINTER(u32)]
    L.suma_map_download.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.suma_map_upload.argtypes = [vp, vp, u32, u32]
    L.suma_map_download_index_map.argtypes = [vp, vp]
    L.suma_map_download_radius_conf.argtypes = [vp, vp]
    L.suma_map_download_poses.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.suma_map_download_integrated.argtypes = [vp, vp]
    L.suma_map_counts.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), vp]
    L.suma_map_cache_stats.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.suma_map_download_cached_tile.argtypes = [vp, C.c_int32, C.c_int32, vp, u32, C.POINTER(u32)]
    L.suma_icp_set_iteration.argtypes = [vp, u32]
    L.suma_pipeline_host_entry_times.argtypes = [vp, vp, C.c_int]
    L.suma_loop_closure_verify.argtypes = [vp, vp, vp, vp, u32, vp, f32, f32, f32, C.POINTER(LoopResult)]
    L.suma_loop_closure_verify_serial.argtypes = [vp, vp, vp, vp, u32, vp, f32, f32, f32, C.POINTER(LoopResult)]
    L.suma_pipeline_create.argtypes = [C.POINTER(SumaParams), C.c_int, pp]
    L.suma_pipeline_destroy.argtypes = [vp]
    L.suma_pipeline_destroy.restype = None
    L.suma_pipeline_ctx.restype = vp
    L.suma_pipeline_ctx.argtypes = [vp]
    L.suma_pipeline_process_scan.argtypes = [vp, vp, vp, vp, u32, i32]
    L.suma_pipeline_process_scan_device.argtypes = [vp, vp, vp, vp, u32, i32]
    L.suma_pipeline_pose.argtypes = [vp, vp]
    L.suma_pipeline_begin_scan.argtypes = [vp, vp, vp, vp, u32]
    L.suma_pipeline_begin_scan_device.argtypes = [vp, vp, vp, vp, u32]
    L.suma_pipeline_begin_prefetched.argtypes = [vp]
    L.suma_pipeline_update_pose.argtypes = [vp, i32]
    L.suma_pipeline_update_map.argtypes = [vp]
    L.suma_pipeline_integrate_loop_closures.argtypes = [vp, vp, u32, vp]
    L.suma_pipeline_set_pose_old.argtypes = [vp, vp]
    L.suma_pipeline_get_pose.argtypes = [vp, C.c_int, vp]
    L.suma_pipeline_result_new.argtypes = [vp, C.POINTER(IcpStats)]
    L.suma_pipeline_verify_loop_closure.argtypes = [vp, vp, vp, u32, f32, f32, C.POINTER(LoopResult)]
    L.suma_pipeline_track_loop_closure.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.POINTER(LoopTrack)]
    L.suma_loop_closure_track.argtypes = [vp, vp, vp, vp, vp, f32, C.c_double, C.c_double, C.c_double, C.POINTER(LoopTrack)]
    L.suma_se3_log.argtypes = [vp, vp]
    L.suma_se3_log.restype = None
    L.suma_pipeline_reset.argtypes = [vp]
    L.suma_pipeline_minimize_hypotheses.argtypes = [vp, vp, u32, i32, vp, vp]
    L.suma_pipeline_apply_increment.argtypes = [vp, vp]
    L.suma_run_sequences.argtypes = [C.POINTER(SumaParams), C.c_int, C.POINTER(SequenceJob), u32, u32, i32,
                                     C.POINTER(SequenceResult)]
    L.suma_run_hypotheses.argtypes = [C.POINTER(SumaParams), C.c_int, C.POINTER(HypothesisJob), i32, EXCHANGE_FN, vp, vp,
                                      vp, C.c_char_p]
    L.suma_pipeline_run_scans.argtypes = [vp, C.POINTER(SequenceJob), i32, C.POINTER(u32), vp]
    L.suma_pipeline_last_increment.argtypes = [vp, vp]
    L.suma_pipeline_last_stats.argtypes = [vp, C.POINTER(IcpStats)]
    L.suma_pipeline_timestamp.restype = u32
    L.suma_pipeline_timestamp.argtypes = [vp]
    L.suma_pipeline_track_loss.restype = u32
    L.suma_pipeline_track_loss.argtypes = [vp]
    L.suma_pipeline_frame.restype = vp
    L.suma_pipeline_frame.argtypes = [vp, C.c_int]
    L.suma_device_alloc.argtypes = [vp, C.c_uint64, pp]
    L.suma_device_free.argtypes = [vp, vp]
    L.suma_device_upload.argtypes = [vp, vp, vp, C.c_uint64]
    L.suma_profile_enable.argtypes = [vp, C.c_int]
    L.suma_profile_reset.argtypes = [vp]
    L.suma_profile_get.argtypes = [vp, C.POINTER(KernelTime), u32]
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _cm(T, dtype):
    """row-major numpy 4x4 -> column-major (Eigen) buffer"""
    return np.ascontiguousarray(np.asarray(T, dtype=dtype).reshape(4, 4).T)


class Context:
    """One HIP device + stream + all device-resident state of the hot path (suma_ctx)."""

    def __init__(self, params: SumaParams, device: int = 0, handle=None, owner=None):
        self.L = lib()
        self.params = params
        self._owner = owner
        if handle is None:
            h = C.c_void_p()
            rc = self.L.suma_ctx_create(C.byref(params), device, C.byref(h))
            if rc != 0:
                raise SumaError(f"suma_ctx_create failed ({rc}): {self.L.suma_last_error(None).decode()}")
            self.h, self.owned = h, True
        else:
            self.h, self.owned = handle, False

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            raise SumaError(f"{what} failed ({rc}): {self.L.suma_last_error(self.h).decode()}")

    def set_params(self, params: SumaParams):
        self.check(self.L.suma_set_params(self.h, C.byref(params)), "suma_set_params")
        self.params = params

    def synchronize(self):
        self.check(self.L.suma_synchronize(self.h), "suma_synchronize")

    @property
    def stream(self) -> int:
        return int(self.L.suma_ctx_stream(self.h) or 0)

    # device scratch for resident scans
    def device_array(self, host: np.ndarray) -> int:
        host = np.ascontiguousarray(host)
        p = C.c_void_p()
        self.check(self.L.suma_device_alloc(self.h, host.nbytes, C.byref(p)), "suma_device_alloc")
        self.check(self.L.suma_device_upload(self.h, p, _ptr(host), host.nbytes), "suma_device_upload")
        return p.value

    def device_download(self, d_ptr: int, nbytes: int) -> np.ndarray:
        """copy `nbytes` from a device address (e.g. an exported viewer buffer) to the host"""
        out = np.empty(nbytes, dtype=np.uint8)
        self.check(self.L.suma_device_download(self.h, _ptr(out), C.c_void_p(d_ptr), nbytes), "suma_device_download")
        return out

    def device_free(self, p: int):
        self.check(self.L.suma_device_free(self.h, C.c_void_p(p)), "suma_device_free")

    # profiling
    def profile(self, on):
        """0 / False: off, 1 / True: every kernel group, 2: only the Gauss-Newton chain (two events per scan)"""
        self.check(self.L.suma_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(self.L.suma_profile_reset(self.h))

    def profile_get(self):
        buf = (KernelTime * 64)()
        n = self.L.suma_profile_get(self.h, buf, 64)
        if n < 0:
            self.check(n, "suma_profile_get")
        return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                     bytes=float(buf[i].bytes)) for i in range(min(n, 64))]

    def close(self):
        if getattr(self, "owned", False) and self.h:
            self.L.suma_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Frame:
    """Frame.h:21-79: vertex / normal / semantic maps (H x W x 4 float32, row 0 = lowest beam) in HBM."""

    def __init__(self, ctx: Context, width: int, height: int, handle=None):
        self.ctx, self.width, self.height = ctx, width, height
        if handle is None:
            h = C.c_void_p()
            ctx.check(ctx.L.suma_frame_create(ctx.h, width, height, C.byref(h)), "suma_frame_create")
            self.h, self.owned = h, True
        else:
            self.h, self.owned = C.c_void_p(handle), False

    def download(self, which: int) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self.ctx.check(self.ctx.L.suma_frame_download(self.ctx.h, self.h, which, _ptr(out)), "suma_frame_download")
        return out

    def upload(self, which: int, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(self.height, self.width, 4)
        self.ctx.check(self.ctx.L.suma_frame_upload(self.ctx.h, self.h, which, _ptr(data)), "suma_frame_upload")

    def set(self, vertex, normal, semantic):
        self.upload(0, vertex)
        self.upload(1, normal)
        self.upload(2, semantic)

    @property
    def vertex(self):
        return self.download(0)

    @property
    def normal(self):
        return self.download(1)

    @property
    def semantic(self):
        return self.download(2)

    def swap(self, other: "Frame"):
        """exchange contents with `other` (the shared_ptr swaps of SurfelMapping.cpp:323-331), O(1)"""
        self.ctx.check(self.ctx.L.suma_frame_swap(self.ctx.h, self.h, other.h), "suma_frame_swap")

    def export(self, which: int):
        """viewer feed: (device address, width, height, row_bytes) of one map"""
        p, w, h, rb = C.c_void_p(), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_frame_export(self.ctx.h, self.h, which, C.byref(p), C.byref(w), C.byref(h),
                                                    C.byref(rb)), "suma_frame_export")
        return p.value, w.value, h.value, rb.value

    def touch(self):
        """the maps were written through an exported device pointer: tell the library (suma_frame_touch)"""
        self.ctx.check(self.ctx.L.suma_frame_touch(self.ctx.h, self.h), "suma_frame_touch")

    def copy(self, other: "Frame"):
        """Frame::copy (Frame.h:49-61)"""
        self.ctx.check(self.ctx.L.suma_frame_copy(self.ctx.h, self.h, other.h), "suma_frame_copy")

    def __del__(self):
        try:
            if getattr(self, "owned", False) and self.h and self.ctx.h:
                self.ctx.L.suma_frame_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Preprocessing:
    """Preprocessing.h:47-58"""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def process(self, points, frame: Frame, labels, probs, timestamp: int) -> Frame:
        points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
        labels = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        probs = None if probs is None else np.ascontiguousarray(probs, dtype=np.float32)
        c = self.ctx
        c.check(c.L.suma_preprocess(c.h, _ptr(points), _ptr(labels), _ptr(probs), points.shape[0], timestamp, frame.h),
                "suma_preprocess")
        return frame


class Frame2Model:
    """Objective.h:14-82 as implemented by Frame2Model.h:28-73.

    Like the reference's object, an instance owns its parameters (icp-max-distance, icp-max-angle, weighting,
    factor, bilinear_sampling; Frame2Model::updateParameters, Frame2Model.cpp:65-110) and its frame pair, and sends
    both to the device before every launch -- two instances with different gates on one Context (objective_ and
    recovery_ = Frame2Model(fallback_params), SurfelMapping.cpp:87-94) do not disturb each other."""
    num_parameters = 6
    _PARAM_NAMES = {"icp-max-distance": "icp_max_distance", "icp-max-angle": "icp_max_angle", "factor": "factor",
                    "bilinear_sampling": "bilinear_sampling"}
    _WEIGHTING = {"none": 0, "huber": 1, "turkey": 2, "stability": 3}

    def __init__(self, ctx: Context, params: SumaParams = None):
        self.ctx = ctx
        p = ctx.params if params is None else params
        self.objective = IcpObjective(p.icp_max_distance, p.icp_max_angle, p.weight_function, p.factor,
                                      p.bilinear_sampling)
        self._current = self._last = None
        self._pose = np.eye(4)
        self._iteration = 0
        self.stats = IcpStats()
        self.acc = np.zeros(ACC_WORDS, dtype=np.int64)

    def setParameter(self, name: str, value):
        """Objective::setParameter(const rv::Parameter&) -> Frame2Model::setParameter (Frame2Model.cpp:112-115)"""
        if name == "weighting":
            self.objective.weight_function = self._WEIGHTING[value]
        elif name in self._PARAM_NAMES:
            setattr(self.objective, self._PARAM_NAMES[name], value)
        # other keys are stored by the reference and never read by this objective

    def setData(self, current: Frame, last: Frame):
        self._current, self._last = current, last  # keep alive
        self._iteration = 0

    def _bind(self):
        """this object's frames and parameters become the ones the next launch uses"""
        if self._current is None:
            raise SumaError("Frame2Model::setData has not been called")
        c = self.ctx
        c.check(c.L.suma_icp_set_data(c.h, self._current.h, self._last.h), "suma_icp_set_data")
        c.check(c.L.suma_icp_set_objective(c.h, C.byref(self.objective)), "suma_icp_set_objective")

    def initialize(self, pose):
        """Objective::initialize (Objective.h:58): the pose and nothing else -- iteration_ is reset by setData only"""
        self._pose = np.asarray(pose, dtype=np.float64).copy()

    def pose(self):
        return self._pose

    def jacobianProducts(self):
        """returns (F, JtJ[6,6], Jtf[6]); updates inlier()/outlier()/valid()/invalid()"""
        JtJ = np.zeros((6, 6), dtype=np.float64)
        Jtr = np.zeros(6, dtype=np.float64)
        pose = _cm(self._pose, np.float64)
        c = self.ctx
        self._bind()
        c.check(c.L.suma_icp_jacobian_products(c.h, _ptr(pose), self._iteration, _ptr(JtJ), _ptr(Jtr), _ptr(self.acc),
                                               C.byref(self.stats)), "suma_icp_jacobian_products")
        return self.stats.error, JtJ.T.copy(), Jtr

    def increment(self, delta):
        """Objective::increment (Objective.h:45-48): pose_ = SE3::exp(delta) * pose_ (host side, numpy)"""
        self._pose = se3_exp(delta) @ self._pose
        self._iteration += 1

    def inlier(self):
        return self.stats.inlier

    def outlier(self):
        return self.stats.outlier

    def valid(self):
        return self.stats.valid

    def invalid(self):
        return self.stats.invalid

    def inlier_residual(self):
        return self.stats.inlier_residual


class LieGaussNewton:
    """LieGaussNewton.h:25-76: the loop itself runs on the device (one readback per minimize)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._pose = np.eye(4)
        self._history = np.zeros((0, 4, 4))
        self.stats = IcpStats()

    def minimize(self, objective: Frame2Model, T0, history_cap: int = 64) -> int:
        T0 = _cm(T0, np.float64)
        T = np.zeros((4, 4), dtype=np.float64)
        nh = C.c_uint32(0)
        c = self.ctx
        objective._bind()
        # Frame2Model::iteration_ runs on across minimisations on one setData (SurfelMapping.cpp:693-700)
        c.check(c.L.suma_icp_set_iteration(c.h, objective._iteration), "suma_icp_set_iteration")
        # the pose history stays on the device until history() asks for it (suma_icp_history)
        c.check(c.L.suma_icp_minimize(c.h, _ptr(T0), _ptr(T), None, 0, C.byref(nh), C.byref(self.stats)), "suma_icp_minimize")
        self._pose = T.T.copy()
        objective._pose = self._pose
        objective.stats = self.stats
        objective._iteration += self.stats.iterations + (1 if self.stats.converged else 0)  # one increment per step
        self._history, self._history_cap = None, history_cap
        self._history_seq = c.L.suma_icp_history_sequence(c.h)
        return 0

    def minimize_batch(self, T0s, objective: Frame2Model = None):
        """n_hyp minimisations of the same frame pair (SurfelMapping.cpp:662-779 pattern)"""
        T0s = np.ascontiguousarray(np.asarray(T0s, dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1))
        n = T0s.shape[0]
        out = np.zeros((n, 4, 4), dtype=np.float64)
        stats = (IcpStats * n)()
        c = self.ctx
        if objective is not None:
            objective._bind()
        c.check(c.L.suma_icp_minimize_batch(c.h, _ptr(T0s), n, _ptr(out), stats), "suma_icp_minimize_batch")
        return out.transpose(0, 2, 1).copy(), [stats[i].as_dict() for i in range(n)]

    def pose(self):
        return self._pose

    def history(self):
        if self._history is None:
            if self.ctx.L.suma_icp_history_sequence(self.ctx.h) != self._history_seq:
                raise RuntimeError("LieGaussNewton.history: a later minimisation on this context has overwritten the "
                                   "device-side history; call history() before it")
            cap = self._history_cap
            hist = np.zeros((max(cap, 1), 4, 4), dtype=np.float64)
            nh = C.c_uint32(0)
            self.ctx.check(self.ctx.L.suma_icp_history(self.ctx.h, _ptr(hist), cap, C.byref(nh)), "suma_icp_history")
            self._history = hist[:min(nh.value, cap)].transpose(0, 2, 1).copy()
        return self._history

    def iterationCount(self):
        return self.stats.iterations

    def information(self):
        """LieGaussNewton::information() (LieGaussNewton.cpp:75,103-105): J^T W J of the last step"""
        out = np.zeros((6, 6), dtype=np.float64)
        self.ctx.check(self.ctx.L.suma_icp_information(self.ctx.h, _ptr(out)), "suma_icp_information")
        return out.T.copy()

    @staticmethod
    def reason(errorno: int) -> str:
        """LieGaussNewton::reason (LieGaussNewton.cpp:110-115)"""
        return {-1: "Maximum number of iterations reached.", -2: "Diverging."}.get(errorno, "no error")


def se3_exp(x):
    """SE3::exp (lie_algebra.cpp:4-34) on the host in numpy, for Objective::increment of the mirror classes"""
    x = np.asarray(x, dtype=np.float64)
    T = np.eye(4)
    v, o = x[:3], x[3:]
    theta = float(np.sqrt(o @ o))
    if theta > 1e-10:
        K = np.array([[0, -o[2], o[1]], [o[2], 0, -o[0]], [-o[1], o[0], 0]])
        K2 = K @ K
        T[:3, :3] = np.eye(3) + np.sin(theta) / theta * K + (1 - np.cos(theta)) / theta ** 2 * K2
        V = np.eye(3) + (1 - np.cos(theta)) / theta ** 2 * K + (theta - np.sin(theta)) / theta ** 3 * K2
        T[:3, 3] = V @ v
    else:
        T[:3, 3] = v
    return T


class SurfelMap:
    """SurfelMap.h:36-78"""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def reset(self):
        self.ctx.check(self.ctx.L.suma_map_reset(self.ctx.h), "suma_map_reset")

    def update(self, pose, frame: Frame):
        p = _cm(pose, np.float32)
        self.ctx.check(self.ctx.L.suma_map_update(self.ctx.h, _ptr(p), frame.h), "suma_map_update")

    def render(self, pose_old, pose_new, frame: Frame, confidence_threshold: float):
        po, pn = _cm(pose_old, np.float32), _cm(pose_new, np.float32)
        self.ctx.check(self.ctx.L.suma_map_render(self.ctx.h, _ptr(po), _ptr(pn), confidence_threshold, frame.h),
                       "suma_map_render")
        return frame

    def render_active(self, pose, confidence_threshold: float):
        p = _cm(pose, np.float32)
        self.ctx.check(self.ctx.L.suma_map_render_active(self.ctx.h, _ptr(p), confidence_threshold))

    def render_inactive(self, pose, confidence_threshold: float):
        p = _cm(pose, np.float32)
        self.ctx.check(self.ctx.L.suma_map_render_inactive(self.ctx.h, _ptr(p), confidence_threshold))

    def render_composed(self, pose_old, pose_new, confidence_threshold: float):
        po, pn = _cm(pose_old, np.float32), _cm(pose_new, np.float32)
        self.ctx.check(self.ctx.L.suma_map_render_composed(self.ctx.h, _ptr(po), _ptr(pn), confidence_threshold))

    def _frame(self, which):
        p = self.ctx.params
        return Frame(self.ctx, p.model_width, p.model_height, handle=self.ctx.L.suma_map_frame(self.ctx.h, which))

    def oldMapFrame(self):
        return self._frame(0)

    def newMapFrame(self):
        return self._frame(1)

    def composedFrame(self):
        return self._frame(2)

    def updatePoses(self, poses):
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))
        self.ctx.check(self.ctx.L.suma_map_update_poses(self.ctx.h, _ptr(poses), poses.shape[0]))

    def size(self) -> int:
        n = C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_size(self.ctx.h, C.byref(n)), "suma_map_size")
        return n.value

    def timestamp(self) -> int:
        t = C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_timestamp(self.ctx.h, C.byref(t)))
        return t.value

    def getAllSurfels(self) -> np.ndarray:
        n = self.size()
        out = np.zeros(n, dtype=SURFEL_DTYPE)
        got = C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_download(self.ctx.h, _ptr(out) if n else None, n, C.byref(got)))
        return out

    def getModelSurfels(self):
        """SurfelMap::getModelSurfels (SurfelMap.h:67) / the VBO SurfelMap::draw reads: (device address, count) of the
        active map, 64-byte records; valid until the next update / upload / reset"""
        p, n = C.c_void_p(), C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_export_surfels(self.ctx.h, C.byref(p), C.byref(n)), "suma_map_export_surfels")
        return p.value, n.value

    def getDataSurfels(self):
        """SurfelMap::getDataSurfels (SurfelMap.h:64): (device address, first, count) -- the new surfels of the last
        update that survived the active-area copy, stored as the tail of the active map"""
        p, f, n = C.c_void_p(), C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_export_data_surfels(self.ctx.h, C.byref(p), C.byref(f), C.byref(n)),
                       "suma_map_export_data_surfels")
        return p.value, f.value, n.value

    def upload(self, surfels: np.ndarray, timestamp: int):
        surfels = np.ascontiguousarray(surfels, dtype=SURFEL_DTYPE)
        self.ctx.check(self.ctx.L.suma_map_upload(self.ctx.h, _ptr(surfels), surfels.shape[0], timestamp))

    # intermediates of the last update (parity tests)
    def index_map(self):
        p = self.ctx.params
        out = np.zeros((p.data_height, p.data_width), dtype=np.uint32)
        self.ctx.check(self.ctx.L.suma_map_download_index_map(self.ctx.h, _ptr(out)))
        return out

    def poses(self):
        """the pose table (SurfelMap::poses_): [timestamp, 4, 4] float32, one pose per integrated scan"""
        t = C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_timestamp(self.ctx.h, C.byref(t)))
        out = np.zeros((max(t.value, 1), 16), dtype=np.float32)
        n = C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_download_poses(self.ctx.h, _ptr(out), t.value, C.byref(n)), "suma_map_download_poses")
        return out[:min(n.value, t.value)].reshape(-1, 4, 4).transpose(0, 2, 1).copy()

    def radius_conf(self):
        p = self.ctx.params
        out = np.zeros((p.data_height, p.data_width, 4), dtype=np.float32)
        self.ctx.check(self.ctx.L.suma_map_download_radius_conf(self.ctx.h, _ptr(out)))
        return out

    def integrated(self):
        p = self.ctx.params
        out = np.zeros((p.data_height, p.data_width), dtype=np.uint8)
        self.ctx.check(self.ctx.L.suma_map_download_integrated(self.ctx.h, _ptr(out)))
        return out

    def cache_stats(self):
        """(surfels allocated from the submap cache arena, its capacity, compactions so far)"""
        a, b, cc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_cache_stats(self.ctx.h, C.byref(a), C.byref(b), C.byref(cc)))
        return a.value, b.value, cc.value

    def cached_tile(self, i: int, j: int) -> np.ndarray:
        """submapCache_(i, j).surfels (SurfelMap.h:186): the records parked for one tile, (n, 16) float32"""
        n = C.c_uint32(0)
        self.ctx.check(self.ctx.L.suma_map_download_cached_tile(self.ctx.h, i, j, None, 0, C.byref(n)))
        out = np.zeros((n.value, 16), dtype=np.float32)
        if n.value:
            self.ctx.check(self.ctx.L.suma_map_download_cached_tile(self.ctx.h, i, j, _ptr(out), n.value, C.byref(n)))
        return out

    def counts(self):
        """(S' survivors of K9, D new surfels of K10, surfels parked in submap caches, submap origin)"""
        a, b, cc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        ij = np.zeros(2, dtype=np.int32)
        self.ctx.check(self.ctx.L.suma_map_counts(self.ctx.h, C.byref(a), C.byref(b), C.byref(cc), _ptr(ij)))
        return a.value, b.value, cc.value, (int(ij[0]), int(ij[1]))


def loop_closure_verify(ctx: Context, current: Frame, pose_prior, initializations, pose_new, conf_threshold: float,
                        min_valid_ratio: float = 0.2, max_outlier_ratio: float = 0.85, serial: bool = False):
    """device side of SurfelMapping::checkLoopClosure (SurfelMapping.cpp:662-757); returns one dict per guess.
    The guesses run as one batched Gauss-Newton chain; serial=True is the reference's one-by-one sequencing (same bits)."""
    prior = _cm(pose_prior, np.float64)
    inits = np.ascontiguousarray(np.asarray(initializations, dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1))
    pn = _cm(pose_new, np.float32)
    n = inits.shape[0]
    res = (LoopResult * n)()
    fn = ctx.L.suma_loop_closure_verify_serial if serial else ctx.L.suma_loop_closure_verify
    ctx.check(fn(ctx.h, current.h, _ptr(prior), _ptr(inits), n, _ptr(pn), conf_threshold, min_valid_ratio,
                 max_outlier_ratio, res), "suma_loop_closure_verify")
    return _loop_results(res, n)


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


def se3_log(T):
    """SE3::log (lie_algebra.cpp:36-71) as the library computes it on the host"""
    Tc = _cm(T, np.float64)
    x = np.zeros(6)
    lib().suma_se3_log(_ptr(Tc), _ptr(x))
    return x


def loop_closure_track(ctx, current, last_pose_old, last_increment, pose_new, conf_threshold, min_valid_ratio=0.2,
                       max_outlier_ratio=0.85, max_increment_difference=0.1):
    """device side of SurfelMapping::checkLoopClosure part 1 (SurfelMapping.cpp:546-574)"""
    r = LoopTrack()
    a, b, pn = _cm(last_pose_old, np.float64), _cm(last_increment, np.float64), _cm(pose_new, np.float32)
    ctx.check(ctx.L.suma_loop_closure_track(ctx.h, current.h, _ptr(a), _ptr(b), _ptr(pn), conf_threshold, min_valid_ratio,
                                            max_outlier_ratio, max_increment_difference, C.byref(r)),
              "suma_loop_closure_track")
    return _loop_track(r)


def _scan_refs(scans, on_device):
    """list of (points, labels, probs[, n]) -> (ScanRef array, keep-alive list).  Host scans: numpy arrays; device
    scans: (d_points, d_labels, d_probs, n) addresses from Context.device_array."""
    refs = (ScanRef * max(1, len(scans)))()
    keep = []
    for k, sc in enumerate(scans):
        if on_device:
            refs[k] = ScanRef(int(sc[0]), int(sc[1]) if sc[1] else None, int(sc[2]) if sc[2] else None, int(sc[3]))
        else:
            pts = np.ascontiguousarray(sc[0], dtype=np.float32).reshape(-1, 4)
            lab = None if sc[1] is None else np.ascontiguousarray(sc[1], dtype=np.float32)
            prob = None if sc[2] is None else np.ascontiguousarray(sc[2], dtype=np.float32)
            keep.append((pts, lab, prob))
            refs[k] = ScanRef(pts.ctypes.data, None if lab is None else lab.ctypes.data,
                              None if prob is None else prob.ctypes.data, pts.shape[0])
    return refs, keep


def run_sequences_native(params: SumaParams, sequences, device: int = 0, fixed_iterations: int = 0,
                         max_concurrent: int = 4, on_device: bool = False):
    """BASELINE configs[3] on one GPU: suma_run_sequences (include/suma_runner.h) -- the sequences (lists of scans) in
    the order given, at most max_concurrent at a time, each through a pipeline and a host thread of its own; no
    interpreter between two scans.  Returns one dict per sequence."""
    L = lib()
    n = len(sequences)
    jobs = (SequenceJob * max(1, n))()
    keep = []
    for j, scans in enumerate(sequences):
        refs, ka = _scan_refs(scans, on_device)
        keep.append((refs, ka))
        jobs[j] = SequenceJob(refs, len(scans), 1 if on_device else 0)
    res = (SequenceResult * max(1, n))()
    rc = L.suma_run_sequences(C.byref(params), device, jobs, n, max_concurrent, fixed_iterations, res)
    out = [dict(status=r.status, scans_done=r.scans_done, map_surfels=r.map_surfels, track_loss=r.track_loss,
                end_pose=np.array(r.end_pose[:]).reshape(4, 4).T.copy(), seconds=r.seconds, error=r.error.decode())
           for r in res[:n]]
    if rc != 0:
        raise SumaError(f"suma_run_sequences failed ({rc}): {[o['error'] for o in out if o['status']]}")
    return out


def run_hypotheses_native(params: SumaParams, scans, perturbations, rank: int = 0, world: int = 1, device: int = 0,
                          fixed_iterations: int = 0, exchange=None, on_device: bool = False):
    """BASELINE configs[2]: suma_run_hypotheses -- per scan len(perturbations) Gauss-Newton chains from
    lastIncrement * perturbations[k] (hypothesis k on rank k % world), one exchange per scan (exchange(local) -> the
    element-wise sum over the ranks of the [n_hyp, 18] table), the same winner on every rank, map update with it.
    Returns (poses [n, 4, 4], winners)."""
    L = lib()
    refs, keep = _scan_refs(scans, on_device)
    D = np.ascontiguousarray(np.asarray(perturbations, dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1))
    n_hyp, n = D.shape[0], len(scans)
    job = HypothesisJob(refs, n, 1 if on_device else 0, D.ctypes.data, n_hyp, rank, world)
    poses = np.zeros((max(1, n), 16), dtype=np.float64)
    winners = np.zeros(max(1, n), dtype=np.int32)
    err = C.create_string_buffer(160)
    failure = []

    def _cb(user, local, allp, count):
        try:
            loc = np.ctypeslib.as_array(local, shape=(count,)).reshape(n_hyp, 18).copy()
            tot = np.ascontiguousarray(exchange(loc), dtype=np.float64).reshape(-1)
            np.ctypeslib.as_array(allp, shape=(count,))[:] = tot
            return 0
        except Exception as e:  # noqa: BLE001 -- reported through the return code
            failure.append(repr(e))
            return -2

    cb = EXCHANGE_FN(_cb) if (world > 1 and exchange is not None) else C.cast(None, EXCHANGE_FN)
    rc = L.suma_run_hypotheses(C.byref(params), device, C.byref(job), fixed_iterations, cb, None, _ptr(poses), _ptr(winners),
                               err)
    if rc != 0:
        raise SumaError(f"suma_run_hypotheses failed ({rc}): {err.value.decode()} {failure}")
    return poses[:n].reshape(n, 4, 4).transpose(0, 2, 1).copy(), [int(w) for w in winners[:n]]


class SurfelMapping:
    """SurfelMapping::processScan (SurfelMapping.cpp:175-210).  processScan* run a scan in one call; beginScan /
    updatePose / updateMap are its phases for hosts that run loop closures between them (verifyLoopClosure,
    trackLoopClosure, setPoseOld, integrateLoopClosures).  The candidate search and the pose graph stay with the host."""

    def __init__(self, params: SumaParams, device: int = 0):
        self.L = lib()
        self.params = params
        h = C.c_void_p()
        rc = self.L.suma_pipeline_create(C.byref(params), device, C.byref(h))
        if rc != 0:
            raise SumaError(f"suma_pipeline_create failed ({rc}): {self.L.suma_last_error(None).decode()}")
        self.h = h
        self.ctx = Context(params, handle=C.c_void_p(self.L.suma_pipeline_ctx(h)), owner=self)
        self.map = SurfelMap(self.ctx)
        self._staged = []

    def processScan(self, points, labels=None, probs=None, fixed_iterations: int = 0):
        points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
        labels = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        probs = None if probs is None else np.ascontiguousarray(probs, dtype=np.float32)
        self.ctx.check(self.L.suma_pipeline_process_scan(self.h, _ptr(points), _ptr(labels), _ptr(probs),
                                                         points.shape[0], fixed_iterations),
                       "suma_pipeline_process_scan")

    def runScans(self, scans, on_device: bool, fixed_iterations: int = 0, call_seconds=None) -> int:
        """the caller's loop over processScan in native code (suma_pipeline_run_scans, include/suma_runner.h): the next
        scans of this pipeline's sequence, one C call for all of them -- what the reference's visualizer thread does
        (VisualizerWindow.cpp:636-689).  scans: device tuples (d_points, d_labels, d_probs, n) or host triples.
        A prepared job (prepareScans) can be passed instead, so that no marshalling sits inside a timed region."""
        job = scans if isinstance(scans, tuple) and len(scans) == 3 and isinstance(scans[0], SequenceJob) else self.prepareScans(scans, on_device)
        done = C.c_uint32(0)
        if call_seconds is not None:  # float64 array of at least n_scans entries: host time of every call
            assert call_seconds.dtype == np.float64 and call_seconds.size >= job[0].n_scans
        self.ctx.check(self.L.suma_pipeline_run_scans(self.h, C.byref(job[0]), fixed_iterations, C.byref(done),
                                                      None if call_seconds is None else call_seconds.ctypes.data),
                       "suma_pipeline_run_scans")
        return done.value

    def prepareScans(self, scans, on_device: bool):
        refs, keep = _scan_refs(scans, on_device)
        return SequenceJob(refs, len(scans), 1 if on_device else 0), refs, keep

    def hostEntryTimes(self, reset: bool = False):
        """per-call averages (us) of where the blocking host-vector entry spends the CALLER's time
        (suma_pipeline_host_entry_times)"""
        out = np.zeros(8, dtype=np.float64)
        self.ctx.check(self.L.suma_pipeline_host_entry_times(self.h, _ptr(out), int(reset)), "suma_pipeline_host_entry_times")
        n = max(out[0], 1.0)
        keys = ("call", "slot_wait", "copy", "upload_enqueue", "kernel_enqueue", "result_wait")
        d = {k: round(1e6 * float(v) / n, 1) for k, v in zip(keys, out[1:7])}
        d["calls"] = int(out[0])
        d["copy_threads"] = int(out[7])
        return d

    def processScanDevice(self, d_points: int, d_labels: int, d_probs: int, n: int, fixed_iterations: int = 0):
        """scan already resident in HBM (device addresses from Context.device_array)"""
        self.ctx.check(self.L.suma_pipeline_process_scan_device(self.h, C.c_void_p(d_points), C.c_void_p(d_labels),
                                                                C.c_void_p(d_probs), n, fixed_iterations),
                       "suma_pipeline_process_scan_device")

    def prefetchScan(self, points, labels=None, probs=None):
        """stage a scan (pinned copy + async upload on the ingest thread / copy stream); the arrays are kept alive
        until the matching processPrefetched() returns"""
        points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
        labels = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        probs = None if probs is None else np.ascontiguousarray(probs, dtype=np.float32)
        self._staged.append((points, labels, probs))
        self.ctx.check(self.L.suma_pipeline_prefetch_scan(self.h, _ptr(points), _ptr(labels), _ptr(probs),
                                                          points.shape[0]), "suma_pipeline_prefetch_scan")

    def processPrefetched(self, fixed_iterations: int = 0):
        try:
            self.ctx.check(self.L.suma_pipeline_process_prefetched(self.h, fixed_iterations),
                           "suma_pipeline_process_prefetched")
        finally:
            # the C side has consumed (and freed) the oldest slot whether or not the scan succeeded
            if self._staged:
                self._staged.pop(0)

    def processSequence(self, scans, fixed_iterations: int = 0, on_scan=None):
        """run an iterable of (points, labels, probs) with the upload of scan k+1 overlapping the kernels of scan k
        (what a reader thread feeding SurfelMapping::processScan does in the reference's visualizer loop)"""
        it = iter(scans)
        ahead = 0
        k = 0
        more = True
        while True:
            while more and ahead < 3:  # keep two scans staged beyond the one about to be processed
                nxt = next(it, None)
                if nxt is None:
                    more = False
                    break
                self.prefetchScan(*nxt[:3])
                ahead += 1
            if ahead == 0:
                return k
            self.processPrefetched(fixed_iterations)
            ahead -= 1
            if on_scan is not None:
                on_scan(k, self)
            k += 1

    # ---- the phases of processScan (SurfelMapping.cpp:175-204)
    def beginScan(self, points, labels=None, probs=None):
        points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
        labels = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        probs = None if probs is None else np.ascontiguousarray(probs, dtype=np.float32)
        self.ctx.check(self.L.suma_pipeline_begin_scan(self.h, _ptr(points), _ptr(labels), _ptr(probs), points.shape[0]),
                       "suma_pipeline_begin_scan")

    def beginScanDevice(self, d_points: int, d_labels: int, d_probs: int, n: int):
        self.ctx.check(self.L.suma_pipeline_begin_scan_device(self.h, C.c_void_p(d_points), C.c_void_p(d_labels),
                                                              C.c_void_p(d_probs), n), "suma_pipeline_begin_scan_device")

    def beginPrefetched(self):
        try:
            self.ctx.check(self.L.suma_pipeline_begin_prefetched(self.h), "suma_pipeline_begin_prefetched")
        finally:
            if self._staged:
                self._staged.pop(0)

    def updatePose(self, fixed_iterations: int = 0):
        self.ctx.check(self.L.suma_pipeline_update_pose(self.h, fixed_iterations), "suma_pipeline_update_pose")

    def updateMap(self):
        self.ctx.check(self.L.suma_pipeline_update_map(self.h), "suma_pipeline_update_map")

    def integrateLoopClosures(self, poses, difference):
        """poses: n x 4 x 4 (row-major numpy) optimised poses -> map_->updatePoses; difference: 4 x 4 double"""
        P = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))
        D = _cm(difference, np.float64)
        self.ctx.check(self.L.suma_pipeline_integrate_loop_closures(self.h, _ptr(P), P.shape[0], _ptr(D)),
                       "suma_pipeline_integrate_loop_closures")

    def setPoseOld(self, pose_old):
        T = _cm(pose_old, np.float64)
        self.ctx.check(self.L.suma_pipeline_set_pose_old(self.h, _ptr(T)), "suma_pipeline_set_pose_old")

    def getPose(self, which: int):
        """0 currentPose_, 1 currentPose_old_, 2 currentPose_new_, 3 lastPose_old_, 4 lastPose_"""
        T = np.zeros((4, 4), dtype=np.float64)
        self.ctx.check(self.L.suma_pipeline_get_pose(self.h, which, _ptr(T)), "suma_pipeline_get_pose")
        return T.T.copy()

    def resultNew(self) -> IcpStats:
        st = IcpStats()
        self.ctx.check(self.L.suma_pipeline_result_new(self.h, C.byref(st)), "suma_pipeline_result_new")
        return st

    def verifyLoopClosure(self, pose_prior, initializations, min_valid_ratio=0.2, max_outlier_ratio=0.85):
        n = len(initializations)
        res = (LoopResult * n)()
        prior = _cm(pose_prior, np.float64)
        inits = np.ascontiguousarray(np.stack([_cm(T, np.float64) for T in initializations]))
        self.ctx.check(self.L.suma_pipeline_verify_loop_closure(self.h, _ptr(prior), _ptr(inits), n, min_valid_ratio,
                                                                max_outlier_ratio, res),
                       "suma_pipeline_verify_loop_closure")
        return _loop_results(res, n)

    def trackLoopClosure(self, min_valid_ratio=0.2, max_outlier_ratio=0.85, max_increment_difference=0.1):
        r = LoopTrack()
        self.ctx.check(self.L.suma_pipeline_track_loop_closure(self.h, min_valid_ratio, max_outlier_ratio,
                                                               max_increment_difference, C.byref(r)),
                       "suma_pipeline_track_loop_closure")
        return _loop_track(r)

    def reset(self):
        """SurfelMapping::reset (SurfelMapping.cpp:131-169)"""
        self.ctx.check(self.L.suma_pipeline_reset(self.h), "suma_pipeline_reset")

    def minimizeHypotheses(self, starts, fixed_iterations: int = 0):
        """n Gauss-Newton chains as one batch against the rendered model, between beginScan and applyIncrement"""
        T0 = np.ascontiguousarray(np.asarray(starts, dtype=np.float64).reshape(-1, 4, 4).transpose(0, 2, 1))
        n = T0.shape[0]
        out = np.zeros((n, 4, 4), dtype=np.float64)
        st = (IcpStats * n)()
        self.ctx.check(self.L.suma_pipeline_minimize_hypotheses(self.h, _ptr(T0), n, fixed_iterations, _ptr(out), st),
                       "suma_pipeline_minimize_hypotheses")
        return out.transpose(0, 2, 1).copy(), [s.as_dict() for s in st]

    def applyIncrement(self, increment):
        T = _cm(increment, np.float64)
        self.ctx.check(self.L.suma_pipeline_apply_increment(self.h, _ptr(T)), "suma_pipeline_apply_increment")

    def getCurrentPose(self):
        T = np.zeros((4, 4), dtype=np.float64)
        self.L.suma_pipeline_pose(self.h, _ptr(T))
        return T.T.copy()

    def lastIncrement(self):
        T = np.zeros((4, 4), dtype=np.float64)
        self.L.suma_pipeline_last_increment(self.h, _ptr(T))
        return T.T.copy()

    def lastStats(self) -> IcpStats:
        st = IcpStats()
        self.L.suma_pipeline_last_stats(self.h, C.byref(st))
        return st

    def minimizeStats(self) -> IcpStats:
        """statistics of the scan's minimisation itself (iterations, converged: suma_pipeline_minimize_stats)"""
        st = IcpStats()
        self.L.suma_pipeline_minimize_stats(self.h, C.byref(st))
        return st

    def timestamp(self) -> int:
        return self.L.suma_pipeline_timestamp(self.h)

    def trackLoss(self) -> int:
        """scans on which the frame-to-frame fallback ran (trackLoss_, SurfelMapping.cpp:441)"""
        return self.L.suma_pipeline_track_loss(self.h)

    def frame(self, which: int) -> Frame:
        """0 current data frame, 1 last model frame, 2 current model frame"""
        p = self.params
        w, h = (p.data_width, p.data_height) if which == 0 else (p.model_width, p.model_height)
        return Frame(self.ctx, w, h, handle=self.L.suma_pipeline_frame(self.h, which))

    def close(self):
        if getattr(self, "h", None):
            self.L.suma_pipeline_destroy(self.h)
            self.h = None
            self.ctx.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
