"""ctypes mirrors of include/suma_types.h (the POD types that cross the C-ABI)."""
from __future__ import annotations

import ctypes as C

import numpy as np

u32, i32, f32, f64 = C.c_uint32, C.c_int32, C.c_float, C.c_double

WEIGHT_NONE, WEIGHT_HUBER, WEIGHT_TUKEY, WEIGHT_STABILITY = 0, 1, 2, 3
MAP_VERTEX, MAP_NORMAL, MAP_SEMANTIC = 0, 1, 2
FRAME_OLD, FRAME_NEW, FRAME_COMPOSED = 0, 1, 2
ACC_WORDS = 32
FILTER_SAMPLING_GL_INITIAL, FILTER_SAMPLING_NEAREST = 0, 1
ACC_SCALE = 268435456.0

# numpy view of the 64-byte surfel record (reference src/core/Surfel.h:5-15)
SURFEL_DTYPE = np.dtype([
    ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("radius", "<f4"),
    ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("confidence", "<f4"),
    ("timestamp", "<u4"), ("color", "<f4"), ("weight", "<f4"), ("count", "<f4"),
    ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("w", "<f4"),
])
assert SURFEL_DTYPE.itemsize == 64


class SumaParams(C.Structure):
    """Flattened rv::ParameterList -- field order identical to ``struct suma_params``."""
    _fields_ = [
        ("data_width", u32), ("data_height", u32), ("data_fov_up", f32), ("data_fov_down", f32),
        ("min_depth", f32), ("max_depth", f32),
        ("model_width", u32), ("model_height", u32), ("model_fov_up", f32), ("model_fov_down", f32),
        ("model_min_depth", f32), ("model_max_depth", f32),
        ("max_iterations", u32), ("stopping_threshold", f32), ("delta", f32),
        ("icp_max_distance", f32), ("icp_max_angle", f32), ("weight_function", i32), ("factor", f32),
        ("bilinear_sampling", i32),
        ("initialize_identity", i32), ("fallback_mode", i32), ("fallback_max_distance", f32),
        ("fallback_max_angle", f32),
        ("compose_rendering", i32), ("max_loop_closure_distance", f32),
        ("min_radius", f32), ("max_radius", f32), ("max_angle", f32), ("map_max_distance", f32),
        ("map_max_angle", f32), ("unstable_age", i32), ("confidence_mode", i32), ("confidence_threshold", f32),
        ("p_stable", f32), ("p_prior", f32), ("sigma_angle", f32), ("sigma_distance", f32),
        ("use_stability", i32), ("active_timestamps", i32), ("max_weight", f32), ("weighting_scheme", i32),
        ("averaging_scheme", i32), ("update_always", i32),
        ("submap_dimension", i32), ("submap_extent", f32), ("partial_extraction", i32),
        ("max_surfels", u32), ("max_poses", u32),
        ("label_offset", u32), ("prob_offset", u32), ("cache_surfels", u32),
        ("avg_vertexmap", i32), ("filter_vertexmap", i32), ("use_filtered_vertexmap", i32),
        ("bilateral_sigma_space", f32), ("bilateral_sigma_range", f32), ("filter_sampling", i32),
    ]


class IcpStats(C.Structure):
    _fields_ = [("error", f64), ("inlier_residual", f64), ("valid", u32), ("outlier", u32), ("inlier", u32),
                ("invalid", u32), ("iterations", u32), ("converged", u32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def default_params(**overrides) -> SumaParams:
    """Values of the reference's config/default.xml (same as suma_params_default in suma_types.h)."""
    p = SumaParams(
        data_width=900, data_height=64, data_fov_up=3.0, data_fov_down=-25.0, min_depth=2.0, max_depth=75.0,
        model_width=900, model_height=64, model_fov_up=3.0, model_fov_down=-25.0, model_min_depth=2.0,
        model_max_depth=75.0,
        max_iterations=33, stopping_threshold=1e-4, delta=1e-4,
        icp_max_distance=2.0, icp_max_angle=30.0, weight_function=WEIGHT_HUBER, factor=0.5, bilinear_sampling=1,
        initialize_identity=0, fallback_mode=1, fallback_max_distance=0.5, fallback_max_angle=30.0,
        compose_rendering=1, max_loop_closure_distance=8.0,
        min_radius=0.03, max_radius=1.0, max_angle=90.0, map_max_distance=0.2, map_max_angle=45.0,
        unstable_age=3, confidence_mode=3, confidence_threshold=0.0, p_stable=0.6, p_prior=0.5,
        sigma_angle=1.0, sigma_distance=1.0, use_stability=1, active_timestamps=100, max_weight=20.0,
        weighting_scheme=0, averaging_scheme=0, update_always=0,
        submap_dimension=4, submap_extent=10.0, partial_extraction=1,
        max_surfels=2048 * 2048, max_poses=10000, label_offset=4, prob_offset=5, cache_surfels=0,
        avg_vertexmap=0, filter_vertexmap=0, use_filtered_vertexmap=0, bilateral_sigma_space=0.0,
        bilateral_sigma_range=2.5, filter_sampling=FILTER_SAMPLING_GL_INITIAL,
    )
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise KeyError(f"unknown parameter {k!r}")
        setattr(p, k, v)
    return p


def params_with_size(width: int, height: int = 64, **overrides) -> SumaParams:
    """default.xml with data and model images of ``width x height`` (BASELINE configs use 64x900 / 64x2048)."""
    return default_params(data_width=width, data_height=height, model_width=width, model_height=height, **overrides)
