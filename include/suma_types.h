/*
 * suma_types.h -- plain-old-data types that cross the C-ABI of the MI355X projective-ICP /
 * surfel-fusion core.  Shared by the product library (include/suma_hip.h) and by the CPU
 * oracle (oracle/), which is test infrastructure only.
 *
 * Every field cites the reference parameter / structure it carries.
 */
#ifndef SUMA_TYPES_H_
#define SUMA_TYPES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One RGBA32F texel of a Frame map (reference: glow::GlTextureRectangle RGBA_FLOAT,
 * src/core/Frame.h:29-32).  Row 0 is the bottom row (lowest beam), x = 0 <-> yaw = +pi. */
typedef struct suma_float4 {
  float x, y, z, w;
} suma_float4;

/* 64-byte surfel record, identical layout to reference src/core/Surfel.h:5-15 (and the
 * transform-feedback varyings of src/core/SurfelMap.cpp:38-40). Position / normal live in the
 * sensor frame of the creation scan; world = poses[int(count)] * p. */
typedef struct suma_surfel {
  float x, y, z, radius;
  float nx, ny, nz, confidence; /* log-odds */
  uint32_t timestamp;           /* last update */
  float color, weight, count;   /* packed debug colour, weight, creation timestamp (as float) */
  float r, g, b, w;             /* label/255 x3, label probability */
} suma_surfel;

/* weight functions of Frame2Model (src/core/Frame2Model.cpp:69-80) */
enum { SUMA_WEIGHT_NONE = 0, SUMA_WEIGHT_HUBER = 1, SUMA_WEIGHT_TUKEY = 2, SUMA_WEIGHT_STABILITY = 3 };

/* which map of a frame */
enum { SUMA_MAP_VERTEX = 0, SUMA_MAP_NORMAL = 1, SUMA_MAP_SEMANTIC = 2 };
/* internal frames owned by the surfel map (src/core/SurfelMap.h:59-61) */
enum { SUMA_FRAME_OLD = 0, SUMA_FRAME_NEW = 1, SUMA_FRAME_COMPOSED = 2 };

/* Flattened rv::ParameterList (config/default.xml).  Raw configuration
 * values only -- derived thresholds (cos/sin of angles, log-odds, pixel size) are computed by
 * each implementation exactly as the reference's setParameters() do. */
typedef struct suma_params {
  /* data image: default.xml:7-12 */
  uint32_t data_width, data_height;
  float data_fov_up, data_fov_down; /* degrees, sign as in the XML (3, -25) */
  float min_depth, max_depth;
  /* model image: default.xml:30-35 */
  uint32_t model_width, model_height;
  float model_fov_up, model_fov_down;
  float model_min_depth, model_max_depth;
  /* Gauss-Newton: default.xml:16-18 (LieGaussNewton.cpp:81-91) */
  uint32_t max_iterations;
  float stopping_threshold; /* epsilon */
  float delta;
  /* Frame2Model: default.xml:19-27 (Frame2Model.cpp:65-110) */
  float icp_max_distance;
  float icp_max_angle; /* degrees */
  int32_t weight_function;
  float factor;
  int32_t bilinear_sampling;
  /* SurfelMapping: default.xml:25,42-49 */
  int32_t initialize_identity;
  int32_t fallback_mode;
  float fallback_max_distance;
  float fallback_max_angle;
  /* surfel rendering: default.xml:37-38 */
  int32_t compose_rendering;
  float max_loop_closure_distance;
  /* surfel update: default.xml:50-63 (SurfelMap.cpp:336-457) */
  float min_radius, max_radius;
  float max_angle;        /* degrees */
  float map_max_distance; /* map-max-distance */
  float map_max_angle;    /* map-max-angle, degrees */
  int32_t unstable_age;
  int32_t confidence_mode;
  float confidence_threshold;
  float p_stable, p_prior;
  float sigma_angle, sigma_distance;
  int32_t use_stability;
  int32_t active_timestamps; /* optional key, default 100 */
  float max_weight;          /* optional key, default 20 */
  int32_t weighting_scheme;  /* default 0 */
  int32_t averaging_scheme;  /* default 0 */
  int32_t update_always;     /* default 0 */
  /* submaps: default.xml:65-67 */
  int32_t submap_dimension;
  float submap_extent;
  int32_t partial_extraction;
  /* capacities (reference constants SurfelMap.h:87,205; made parameters here) */
  uint32_t max_surfels; /* reference: 2048*2048 */
  uint32_t max_poses;   /* reference: 10000 */
  /* reference quirk B-1 (Preprocessing.cpp:142-145): point i receives labels[i+4], probs[i+5].
   * label_offset/prob_offset reproduce (4,5) or fix (0,0) it; reads past the end yield 0. */
  uint32_t label_offset, prob_offset;
  /* capacity (in surfels) of the HBM arena that holds the parked submap tiles; the reference keeps
   * them in host RAM (SurfelMap.h:186).  0 = 16 * max_surfels. */
  uint32_t cache_surfels;
  /* optional vertex-map filters of Preprocessing (Preprocessing.cpp:76-90,150-236); config/default.xml leaves
   * `avg_vertexmap` / `filter_vertexmap` out (= off) and holds no `bilateral_sigma_space` at all.
   *   avg_vertexmap          K1 without depth test, additive blending (sums of (x,y,z,1) and of the label texel,
   *                          in point order), then K1b avg_vertexmap.frag divides the vertex sum by its count
   *   filter_vertexmap       K1c bilateral_filter.frag (13x13 range filter); its output replaces the vertex map
   *                          only when use_filtered_vertexmap is set (Preprocessing.cpp:234), as in the reference
   *   filter_sampling        how the two filter shaders see their input texture.  They sample a rectangle
   *                          texture at INTEGER coordinates with no sampler object bound (Preprocessing.cpp:198,
   *                          217), i.e. with the texture's own state, which un-vendored glow sets.
   *                          SUMA_FILTER_SAMPLING_GL_INITIAL = the GL initial state of a rectangle texture
   *                          (LINEAR, CLAMP_TO_EDGE: GL 3.3 core 3.8.15) -- an integer coordinate is a texel
   *                          CORNER, the fetch averages the four texels around it; SUMA_FILTER_SAMPLING_NEAREST
   *                          = texel (x, y) itself. */
  int32_t avg_vertexmap;
  int32_t filter_vertexmap;
  int32_t use_filtered_vertexmap;
  float bilateral_sigma_space; /* no default in default.xml: must be > 0 when filter_vertexmap is set */
  float bilateral_sigma_range; /* default.xml:79 */
  int32_t filter_sampling;
} suma_params;

#define SUMA_FILTER_SAMPLING_GL_INITIAL 0
#define SUMA_FILTER_SAMPLING_NEAREST 1

/* Unpacked row 7 of the reference's 2x8 blend target (Frame2Model.cpp:222-227) */
typedef struct suma_icp_stats {
  double error;           /* F = sum w r^2 over valid pairs (blending[43]) */
  double inlier_residual; /* sum w r^2 over inliers (blending[45]) */
  uint32_t valid;         /* inlier + outlier (blending[42]) */
  uint32_t outlier;       /* blending[44] */
  uint32_t inlier;        /* valid - outlier */
  uint32_t invalid;       /* blending[46] */
  uint32_t iterations;    /* LieGaussNewton::iterationCount() (minimize only) */
  uint32_t converged;     /* 1 if a stopping test fired before max_iterations */
} suma_icp_stats;

/* Fixed-point scale of the order-independent JtJ / Jtr accumulation ("K6"):
 * every per-pixel fp32 term is converted to int64 with round-to-nearest-even at 2^28. */
#define SUMA_ACC_SCALE 268435456.0
#define SUMA_ACC_WORDS 32 /* 21 upper-tri JtJ + 6 Jtr + F + F_inlier + valid + outlier + invalid */

/* Fill a parameter block with the values of the reference's config/default.xml. */
static inline void suma_params_default(suma_params* p) {
  p->data_width = 900;
  p->data_height = 64;
  p->data_fov_up = 3.0f;
  p->data_fov_down = -25.0f;
  p->min_depth = 2.0f;
  p->max_depth = 75.0f;
  p->model_width = 900;
  p->model_height = 64;
  p->model_fov_up = 3.0f;
  p->model_fov_down = -25.0f;
  p->model_min_depth = 2.0f;
  p->model_max_depth = 75.0f;
  p->max_iterations = 33;
  p->stopping_threshold = 0.0001f;
  p->delta = 0.0001f;
  p->icp_max_distance = 2.0f;
  p->icp_max_angle = 30.0f;
  p->weight_function = SUMA_WEIGHT_HUBER;
  p->factor = 0.5f;
  p->bilinear_sampling = 1;
  p->initialize_identity = 0;
  p->fallback_mode = 1;
  p->fallback_max_distance = 0.5f;
  p->fallback_max_angle = 30.0f;
  p->compose_rendering = 1;
  p->max_loop_closure_distance = 8.0f;
  p->min_radius = 0.03f;
  p->max_radius = 1.0f;
  p->max_angle = 90.0f;
  p->map_max_distance = 0.2f;
  p->map_max_angle = 45.0f;
  p->unstable_age = 3;
  p->confidence_mode = 3;
  p->confidence_threshold = 0.0f;
  p->p_stable = 0.6f;
  p->p_prior = 0.5f;
  p->sigma_angle = 1.0f;
  p->sigma_distance = 1.0f;
  p->use_stability = 1;
  p->active_timestamps = 100;
  p->max_weight = 20.0f;
  p->weighting_scheme = 0;
  p->averaging_scheme = 0;
  p->update_always = 0;
  p->submap_dimension = 4;
  p->submap_extent = 10.0f;
  p->partial_extraction = 1;
  p->max_surfels = 2048u * 2048u;
  p->max_poses = 10000;
  p->label_offset = 4;
  p->prob_offset = 5;
  p->cache_surfels = 0;
  p->avg_vertexmap = 0;
  p->filter_vertexmap = 0;
  p->use_filtered_vertexmap = 0;
  p->bilateral_sigma_space = 0.0f;
  p->bilateral_sigma_range = 2.5f;
  p->filter_sampling = SUMA_FILTER_SAMPLING_GL_INITIAL;
}

#ifdef __cplusplus
}
#endif
#endif /* SUMA_TYPES_H_ */
