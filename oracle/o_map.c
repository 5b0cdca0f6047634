/*
 * oracle/o_map.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Restates SurfelMap (reference src/core/SurfelMap.cpp) and its shaders:
 *   K4  render_surfels.vert:42-54 / .geom:76-123 / .frag:19-33   (SurfelMap.cpp:847-1165)
 *   K5  render_compose.frag:26-48                                 (SurfelMap.cpp:911-940)
 *   K7  gen_indexmap.vert:62-81                                   (SurfelMap.cpp:586-604)
 *   K8  init_radiusConf.vert:34-68                                (SurfelMap.cpp:606-619)
 *   K9  update_surfels.vert:140-334 / .geom:30-43 / .frag:9-12    (SurfelMap.cpp:621-644)
 *   K10 gen_surfels.vert:38-52 / gen_surfels.geom:109-145         (SurfelMap.cpp:646-664)
 *   K11 copy_surfels.vert:38-56                                   (SurfelMap.cpp:667-698)
 *   K12 extract_surfels.vert:46-64 + updateActiveSubmaps          (SurfelMap.cpp:708-824)
 *
 * GL rules restated where the shaders rely on the fixed-function pipeline: point clipping,
 * depth test on a 24-bit buffer with in-order tie breaking, transform-feedback ordering, and --
 * for K4 -- a triangle rasteriser.  GL leaves sub-pixel snapping to the implementation; this
 * oracle fixes it to 8 sub-pixel bits with integer edge functions (DESIGN.md "K4"), which makes
 * coverage exactly reproducible on CPU and GPU.
 */
#include "o_ctx.h"

#define O_EXTRACT_CAPACITY 500000u /* SurfelMap.cpp:279 */
#define O_EMPTY (~(uint64_t)0)

static inline float o_deg2rad(float deg) { return (float)((double)deg * M_PI / 180.0); }

typedef struct o_map_consts {
  float pixel_size, log_prior, log_unstable, p_unstable;
  float radconf_angle_thresh, update_angle_thresh;
} o_map_consts;

/* SurfelMap::setParameters, SurfelMap.cpp:336-457 */
static o_map_consts o_consts(const suma_params* p) {
  o_map_consts k;
  float vfov = fabsf(p->data_fov_up) + fabsf(p->data_fov_down);
  float hfov = 360.0f;
  /* SurfelMap.cpp:342-343: rv::Math::deg2rad is double (Math.h:44-47) -> the argument is evaluated in double */
  float vpix = (float)tan(0.5 * ((double)vfov * M_PI / 180.0) / (double)p->data_height);
  float hpix = (float)tan(0.5 * ((double)hfov * M_PI / 180.0) / (double)p->data_width);
  k.pixel_size = of_max(vpix, hpix);
  k.p_unstable = 1.0f - p->p_stable;
  k.log_prior = (float)log((double)p->p_prior / (1.0 - (double)p->p_prior));
  k.log_unstable = (float)log((double)k.p_unstable / (1.0 - (double)k.p_unstable));
  k.radconf_angle_thresh = (float)cos((double)o_deg2rad(p->max_angle));    /* :394-397 */
  /* :407 std::sin(Radians(float)): rv/geometry.h:81-84 evaluates ((float)M_PI / 180.f) * deg in float; sinf */
  k.update_angle_thresh = sinf(((float)M_PI / 180.f) * p->map_max_angle);
  return k;
}

void o_map_alloc(ora_ctx* c) {
  const suma_params* p = &c->p;
  size_t P = (size_t)p->data_width * p->data_height, Pm = (size_t)p->model_width * p->model_height;
  c->surfels = (suma_surfel*)malloc((size_t)p->max_surfels * sizeof(suma_surfel));
  c->updated = (suma_surfel*)malloc((size_t)p->max_surfels * sizeof(suma_surfel));
  c->data_surfels = (suma_surfel*)malloc(2 * P * sizeof(suma_surfel)); /* SurfelMap.cpp:57 */
  c->poses = (float*)malloc((size_t)p->max_poses * 16 * sizeof(float));
  c->poses_inv = (float*)malloc((size_t)p->max_poses * 16 * sizeof(float));
  c->old_frame = ora_frame_create(p->model_width, p->model_height);
  c->new_frame = ora_frame_create(p->model_width, p->model_height);
  c->composed_frame = ora_frame_create(p->model_width, p->model_height);
  c->index_map = (uint32_t*)calloc(P, sizeof(uint32_t));
  c->radius_conf = (suma_float4*)calloc(P, sizeof(suma_float4));
  c->integrated = (uint8_t*)calloc(P, 1);
  c->zbuf_a = (uint64_t*)malloc(Pm * sizeof(uint64_t));
  c->zbuf_b = (uint64_t*)malloc(Pm * sizeof(uint64_t));
  c->zbuf_data = (uint64_t*)malloc(P * sizeof(uint64_t));
  c->caches = NULL;
  c->n_caches = c->cap_caches = 0;
  c->extraction = NULL;
  c->n_extraction = c->cap_extraction = 0;
  ora_map_reset(c);
}

void o_map_free(ora_ctx* c) {
  free(c->scratch_flags);
  c->scratch_flags = NULL;
  c->scratch_flags_cap = 0;
  free(c->surfels);
  free(c->updated);
  free(c->data_surfels);
  free(c->poses);
  free(c->poses_inv);
  ora_frame_destroy(c->old_frame);
  ora_frame_destroy(c->new_frame);
  ora_frame_destroy(c->composed_frame);
  free(c->index_map);
  free(c->radius_conf);
  free(c->integrated);
  free(c->zbuf_a);
  free(c->zbuf_b);
  free(c->zbuf_data);
  for (uint32_t i = 0; i < c->n_caches; ++i) free(c->caches[i].surfels);
  free(c->caches);
  free(c->extraction);
}

static void o_set_pose(ora_ctx* c, uint32_t t, const float* pose) {
  memcpy(c->poses + 16 * (size_t)t, pose, 16 * sizeof(float));
  om4_rigid_inverse(pose, c->poses_inv + 16 * (size_t)t);
}

/* SurfelMap::reset, SurfelMap.cpp:473-482 */
void ora_map_reset(ora_ctx* c) {
  static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  c->n_surfels = c->n_updated = c->n_data = 0;
  c->timestamp = 0;
  for (uint32_t t = 0; t < c->p.max_poses; ++t) o_set_pose(c, t, I);
  for (uint32_t i = 0; i < c->n_caches; ++i) free(c->caches[i].surfels);
  c->n_caches = 0;
  c->n_extraction = 0;
  c->n_extractions_done = 0;
  c->origin_i = c->origin_j = 0;
}

/* SurfelMap::updatePoses, SurfelMap.cpp:485-490 */
void ora_map_update_poses(ora_ctx* c, const float* poses16, uint32_t n) {
  for (uint32_t i = 0; i < n && i < c->p.max_poses; ++i) o_set_pose(c, i, poses16 + 16 * (size_t)i);
}

uint32_t ora_map_size(const ora_ctx* c) { return c->n_surfels; }
uint32_t ora_map_timestamp(const ora_ctx* c) { return c->timestamp; }
const suma_surfel* ora_map_surfels(const ora_ctx* c) { return c->surfels; }
const uint32_t* ora_map_index_map(const ora_ctx* c) { return c->index_map; }
const suma_float4* ora_map_radius_conf(const ora_ctx* c) { return c->radius_conf; }
const uint8_t* ora_map_integrated(const ora_ctx* c) { return c->integrated; }
uint32_t ora_map_last_updated_count(const ora_ctx* c) { return c->n_updated; }
uint32_t ora_map_last_new_count(const ora_ctx* c) { return c->n_data; }
ora_frame* ora_map_frame(ora_ctx* c, int which) {
  return which == SUMA_FRAME_OLD ? c->old_frame : (which == SUMA_FRAME_NEW ? c->new_frame : c->composed_frame);
}
void ora_map_upload(ora_ctx* c, const suma_surfel* s, uint32_t n, uint32_t timestamp) {
  if (n > c->p.max_surfels) n = c->p.max_surfels;
  memcpy(c->surfels, s, (size_t)n * sizeof(suma_surfel));
  c->n_surfels = n;
  c->timestamp = timestamp;
}
const suma_surfel* ora_map_updated_surfels(const ora_ctx* c) { return c->updated; }
const suma_surfel* ora_map_data_surfels(const ora_ctx* c) { return c->data_surfels; }
const float* ora_map_poses(const ora_ctx* c) { return c->poses; }
uint32_t ora_map_pending_extractions(const ora_ctx* c) { return c->n_extraction; }
/* tile (i, j) of the submap cache: records and count (NULL / 0 when the tile was never extracted) */
const suma_surfel* ora_map_cache_tile(const ora_ctx* c, int32_t i, int32_t j, uint32_t* n) {
  for (uint32_t k = 0; k < c->n_caches; ++k)
    if (c->caches[k].i == i && c->caches[k].j == j) {
      *n = c->caches[k].n;
      return c->caches[k].surfels;
    }
  *n = 0;
  return NULL;
}
uint32_t ora_map_last_extraction(const ora_ctx* c, int32_t* ij) {
  ij[0] = c->last_extract_i;
  ij[1] = c->last_extract_j;
  return c->n_extractions_done;
}
uint32_t ora_map_cached_surfels(const ora_ctx* c) {
  uint32_t s = 0;
  for (uint32_t i = 0; i < c->n_caches; ++i) s += c->caches[i].n;
  return s;
}
void ora_map_submap_origin(const ora_ctx* c, int32_t* ij) {
  ij[0] = c->origin_i;
  ij[1] = c->origin_j;
}

/* ------------------------------------------------------------------------------------------
 * K4: surfel rendering
 * ---------------------------------------------------------------------------------------- */

typedef struct o_rvtx {
  int64_t X, Y; /* window coordinates in 1/256 pixel */
  float z, tu, tv;
} o_rvtx;

static inline int64_t o_edge(const o_rvtx* a, const o_rvtx* b, int64_t px, int64_t py) {
  return (b->X - a->X) * (py - a->Y) - (b->Y - a->Y) * (px - a->X);
}
/* tie rule for a pixel centre exactly on an edge: antisymmetric in the edge direction, so a
 * pixel on the diagonal shared by the two strip triangles is produced exactly once */
static inline int o_owns(const o_rvtx* s, const o_rvtx* t) {
  int64_t dx = t->X - s->X, dy = t->Y - s->Y;
  return dy > 0 || (dy == 0 && dx < 0);
}
static inline int64_t o_floordiv256(int64_t v) { return v >> 8; } /* arithmetic shift = floor */

enum { O_TIE_LOW_INDEX = 0, O_TIE_HIGH_INDEX_OLD = 1, O_TIE_HIGH_INDEX_NEW = 2 };

static inline uint64_t o_render_key(uint32_t z24, uint32_t id, int tie) {
  /* GL_LESS, in-order: equal depth keeps the earlier primitive (lower id).
   * GL_LEQUAL (render_composed, SurfelMap.cpp:1126): equal depth takes the later primitive;
   * the "new" pass is drawn after the "old" pass into the same depth buffer. */
  if (tie == O_TIE_LOW_INDEX) return ((uint64_t)z24 << 32) | id;
  uint64_t pass = (tie == O_TIE_HIGH_INDEX_OLD) ? 1u : 0u;
  return ((uint64_t)z24 << 33) | (pass << 32) | (uint64_t)(0xffffffffu - id);
}

static void o_raster_tri(o_rvtx A, o_rvtx B, o_rvtx C, int32_t W, int32_t H, uint64_t* zbuf, uint32_t id, int tie) {
  int64_t area = o_edge(&A, &B, C.X, C.Y);
  if (area == 0) return;
  if (area < 0) {
    o_rvtx t = B;
    B = C;
    C = t;
    area = -area;
  }
  int64_t minX = A.X < B.X ? A.X : B.X, maxX = A.X > B.X ? A.X : B.X;
  int64_t minY = A.Y < B.Y ? A.Y : B.Y, maxY = A.Y > B.Y ? A.Y : B.Y;
  if (C.X < minX) minX = C.X;
  if (C.X > maxX) maxX = C.X;
  if (C.Y < minY) minY = C.Y;
  if (C.Y > maxY) maxY = C.Y;
  int64_t i0 = o_floordiv256(minX - 128 + 255), i1 = o_floordiv256(maxX - 128);
  int64_t j0 = o_floordiv256(minY - 128 + 255), j1 = o_floordiv256(maxY - 128);
  if (i0 < 0) i0 = 0;
  if (j0 < 0) j0 = 0;
  if (i1 > W - 1) i1 = W - 1;
  if (j1 > H - 1) j1 = H - 1;
  const int own0 = o_owns(&B, &C), own1 = o_owns(&C, &A), own2 = o_owns(&A, &B);
  const float fa = (float)area;
  for (int64_t j = j0; j <= j1; ++j) {
    for (int64_t i = i0; i <= i1; ++i) {
      int64_t px = 256 * i + 128, py = 256 * j + 128;
      int64_t w0 = o_edge(&B, &C, px, py), w1 = o_edge(&C, &A, px, py), w2 = o_edge(&A, &B, px, py);
      if (!(w0 > 0 || (w0 == 0 && own0))) continue;
      if (!(w1 > 0 || (w1 == 0 && own1))) continue;
      if (!(w2 > 0 || (w2 == 0 && own2))) continue;
      float b0 = (float)w0 / fa, b1 = (float)w1 / fa, b2 = (float)w2 / fa;
      float tu = (b0 * A.tu + b1 * B.tu) + b2 * C.tu;
      float tv = (b0 * A.tv + b1 * B.tv) + b2 * C.tv;
      if ((tu * tu + tv * tv) > 1.0f) continue; /* render_surfels.frag:22 */
      float z = (b0 * A.z + b1 * B.z) + b2 * C.z;
      if (!(z >= 0.0f && z <= 1.0f)) continue; /* near / far clipping */
      uint64_t key = o_render_key(o_depth24(z), id, tie);
      size_t pix = (size_t)j * (size_t)W + (size_t)i;
      o_zmin(&zbuf[pix], key);
    }
  }
}

/* per-surfel transform shared by K4 / K7: (inv_pose * surfelPose) * v, render_surfels.vert:44-48 */
static inline void o_surfel_to_sensor(const ora_ctx* c, const float* inv_pose, const suma_surfel* s, ov3* p, ov3* n) {
  float M[16];
  om4_mul(inv_pose, c->poses + 16 * (size_t)(int32_t)s->count, M);
  *p = om4_point(M, ov3_make(s->x, s->y, s->z));
  *n = om4_dir(M, ov3_make(s->nx, s->ny, s->nz));
}

/* mode: 0 = old surfels, 1 = new surfels, 2 = all (non-compose mode: threshold 0, "new") */
static int o_render_selects(const ora_ctx* c, const suma_surfel* s, int mode, int32_t thr) {
  int32_t creation = (int32_t)s->count;
  int32_t ts = (int32_t)s->timestamp;
  if (mode == 0) return creation < thr;
  (void)c;
  return (creation >= thr || ts >= thr); /* render_surfels.geom:90-91 */
}

/* render_surfels.vert:42-54 + render_surfels.geom:76-123 for one surfel: gate and the four strip corners in
 * [0,1]^3 (x unwrapped relative to the centre, .geom:67-69).  Returns 1 when the quad is emitted. */
static int o_render_quad(const ora_ctx* c, const ora_proj* q, const float* inv_pose, float conf_threshold, int mode,
                         int32_t thr, const suma_surfel* s, ov3* p_out, ov3* n_out, ov3 pr[4]) {
  ov3 p, n;
  o_surfel_to_sensor(c, inv_pose, s, &p, &n);
  *p_out = p;
  *n_out = n;
  ov3 u = ov3_normalize(ov3_make(n.y - n.z, -n.x, n.x));
  ov3 v = ov3_normalize(ov3_cross(n, u));
  float r = s->radius;
  float lp = ov3_len(p);
  int visible = ov3_dot(n, ov3_divs(ov3_neg(p), lp)) > 0.01f;
  ov3 pp = o_project01(q, p);
  if (!(visible && pp.x >= 0.0f && pp.y >= 0.0f && pp.z >= 0.0f && pp.x < 1.0f && pp.y < 1.0f && pp.z < 1.0f &&
        (!c->p.use_stability || s->confidence > conf_threshold)))
    return 0;
  if (!o_render_selects(c, s, mode, thr)) return 0;
  ov3 ru = ov3_scale(r, u), rv = ov3_scale(r, v);
  ov3 corner[4];
  corner[0] = ov3_sub(ov3_sub(p, ru), rv);
  corner[1] = ov3_sub(ov3_add(p, ru), rv);
  corner[2] = ov3_add(ov3_sub(p, ru), rv);
  corner[3] = ov3_add(ov3_add(p, ru), rv);
  for (int k = 0; k < 4; ++k) {
    pr[k] = o_project01(q, corner[k]);
    /* render_surfels.geom:67-69 seam hack */
    if (pp.x - pr[k].x > 0.5f) pr[k].x += 1.0f;
    if (pr[k].x - pp.x > 0.5f) pr[k].x -= 1.0f;
  }
  return 1;
}

static void o_render_pass(const ora_ctx* c, const float* inv_pose, float conf_threshold, int mode, int32_t thr,
                          uint64_t* zbuf, int tie) {
  const ora_proj q = o_proj_model(&c->p);
  const int32_t W = (int32_t)c->p.model_width, H = (int32_t)c->p.model_height;
#pragma omp parallel for num_threads(c->threads) schedule(dynamic, 1024)
  for (uint32_t i = 0; i < c->n_surfels; ++i) {
    ov3 p, n, pr[4];
    if (!o_render_quad(c, &q, inv_pose, conf_threshold, mode, thr, &c->surfels[i], &p, &n, pr)) continue;
    static const float tcu[4] = {-1.f, 1.f, -1.f, 1.f}, tcv[4] = {-1.f, -1.f, 1.f, 1.f};
    o_rvtx vt[4];
    int bad = 0;
    for (int k = 0; k < 4; ++k) {
      float xw = pr[k].x * q.width, yw = pr[k].y * q.height;
      if (sdm_isnan(xw) || sdm_isnan(yw) || sdm_isnan(pr[k].z)) bad = 1;
      vt[k].X = (int64_t)sdm_floor(xw * 256.0f + 0.5f);
      vt[k].Y = (int64_t)sdm_floor(yw * 256.0f + 0.5f);
      /* gl_Position.z = 2 z01 - 1 (render_surfels.geom:104-117), viewport with depth range [0, 1]: z_w = 0.5 z_ndc + 0.5
       * -- in fp32 this is z01 again for only 84 % of the values (a real GL interpolates and quantises z_w) */
      vt[k].z = 0.5f * (2.0f * pr[k].z - 1.0f) + 0.5f;
      vt[k].tu = tcu[k];
      vt[k].tv = tcv[k];
    }
    if (bad) continue;
    /* triangle strip: (v0,v1,v2), (v2,v1,v3) */
    o_raster_tri(vt[0], vt[1], vt[2], W, H, zbuf, i, tie);
    o_raster_tri(vt[2], vt[1], vt[3], W, H, zbuf, i, tie);
  }
}

/* stage export for tests/test_ref_shaders.py: what K4's vertex + geometry stage produce per surfel, before
 * rasterisation.  mode / thr as in o_render_pass; corners = 4 x (x01, y01, z01); pn = sensor-frame position, normal */
void ora_debug_render_quads(const ora_ctx* c, const float pose[16], float conf_threshold, int mode, int32_t thr,
                            uint8_t* emitted, float* corners, float* pn) {
  const ora_proj q = o_proj_model(&c->p);
  float inv_pose[16];
  om4_rigid_inverse(pose, inv_pose);
  for (uint32_t i = 0; i < c->n_surfels; ++i) {
    ov3 p, n, pr[4];
    memset(pr, 0, sizeof(pr));
    emitted[i] = (uint8_t)o_render_quad(c, &q, inv_pose, conf_threshold, mode, thr, &c->surfels[i], &p, &n, pr);
    for (int k = 0; k < 4; ++k) {
      corners[12 * (size_t)i + 3 * k] = pr[k].x;
      corners[12 * (size_t)i + 3 * k + 1] = pr[k].y;
      corners[12 * (size_t)i + 3 * k + 2] = pr[k].z;
    }
    pn[6 * (size_t)i] = p.x;
    pn[6 * (size_t)i + 1] = p.y;
    pn[6 * (size_t)i + 2] = p.z;
    pn[6 * (size_t)i + 3] = n.x;
    pn[6 * (size_t)i + 4] = n.y;
    pn[6 * (size_t)i + 5] = n.z;
  }
}

static inline uint32_t o_key_id(uint64_t key, int tie) {
  uint32_t low = (uint32_t)(key & 0xffffffffu);
  return tie == O_TIE_LOW_INDEX ? low : (0xffffffffu - low);
}

/* the rasteriser alone, for tests/test_gl_reference.py: the quads are given (n x 4 corners in [0,1]^3, drawn in this
 * order as the strip (v0,v1,v2), (v2,v1,v3) with primitive id ids[k]), a real GL implementation rasterises the same
 * vertices behind a pass-through shader, and the two winner maps are compared.  use_disc = 0: no fragment is
 * discarded; use_depth = 0: no depth test, the last primitive that covers a pixel owns it (ids ascending).
 * winner: W x H, primitive id or -1. */
void ora_debug_raster_quads(int32_t W, int32_t H, const float* corners, const uint32_t* ids, uint32_t n, int use_disc,
                            int use_depth, int64_t* winner) {
  uint64_t* zbuf = (uint64_t*)malloc((size_t)W * (size_t)H * sizeof(uint64_t));
  for (size_t k = 0; k < (size_t)W * (size_t)H; ++k) zbuf[k] = ~0ull;
  static const float tcu[4] = {-1.f, 1.f, -1.f, 1.f}, tcv[4] = {-1.f, -1.f, 1.f, 1.f};
  const int tie = use_depth ? O_TIE_LOW_INDEX : O_TIE_HIGH_INDEX_NEW;
  for (uint32_t q = 0; q < n; ++q) {
    o_rvtx vt[4];
    int bad = 0;
    for (int k = 0; k < 4; ++k) {
      const float* c = corners + 12 * (size_t)q + 3 * k;
      const float xw = c[0] * (float)W, yw = c[1] * (float)H;
      if (sdm_isnan(xw) || sdm_isnan(yw) || sdm_isnan(c[2])) bad = 1;
      vt[k].X = (int64_t)sdm_floor(xw * 256.0f + 0.5f);
      vt[k].Y = (int64_t)sdm_floor(yw * 256.0f + 0.5f);
      vt[k].z = use_depth ? 0.5f * (2.0f * c[2] - 1.0f) + 0.5f : 0.0f;
      vt[k].tu = use_disc ? tcu[k] : 0.0f;
      vt[k].tv = use_disc ? tcv[k] : 0.0f;
    }
    if (bad) continue;
    o_raster_tri(vt[0], vt[1], vt[2], W, H, zbuf, ids[q], tie);
    o_raster_tri(vt[2], vt[1], vt[3], W, H, zbuf, ids[q], tie);
  }
  for (size_t k = 0; k < (size_t)W * (size_t)H; ++k)
    winner[k] = (zbuf[k] == ~0ull) ? -1 : (int64_t)o_key_id(zbuf[k], tie);
  free(zbuf);
}
/* o_depth24 of n window depths */
void ora_debug_depth24(const float* zw, uint32_t n, uint32_t* out) {
  for (uint32_t k = 0; k < n; ++k) out[k] = o_depth24(zw[k]);
}


/* write vertex / normal / semantic of the winning surfel (render_surfels.frag:30-32) */
static void o_render_resolve(const ora_ctx* c, const float* inv_pose_a, const float* inv_pose_b, const uint64_t* zbuf,
                             int tie, suma_float4* vmap, suma_float4* nmap, suma_float4* smap) {
  const size_t Pm = (size_t)c->p.model_width * c->p.model_height;
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (size_t pix = 0; pix < Pm; ++pix) {
    uint64_t key = zbuf[pix];
    if (key == O_EMPTY) {
      if (vmap) vmap[pix] = o_f4(0.f, 0.f, 0.f, 0.f);
      if (nmap) nmap[pix] = o_f4(0.f, 0.f, 0.f, 0.f);
      if (smap) smap[pix] = o_f4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    uint32_t id = o_key_id(key, tie);
    const suma_surfel* s = &c->surfels[id];
    const float* inv_pose = inv_pose_a;
    if (tie != O_TIE_LOW_INDEX && ((key >> 32) & 1u) == 0) inv_pose = inv_pose_b; /* composed: new pass */
    ov3 p, n;
    o_surfel_to_sensor(c, inv_pose, s, &p, &n);
    if (vmap) vmap[pix] = o_f4(p.x, p.y, p.z, 1.0f);
    if (nmap) nmap[pix] = o_f4(n.x, n.y, n.z, 1.0f);
    if (smap) smap[pix] = o_f4(s->r, s->g, s->b, s->w);
  }
}

static void o_clear_zbuf(uint64_t* z, size_t n) {
  for (size_t i = 0; i < n; ++i) z[i] = O_EMPTY;
}

/* SurfelMap::render(pose_old, pose_new, frame, ct), SurfelMap.cpp:847-1021.  The two passes
 * into composedFrame_ (SurfelMap.cpp:893-906) are dead work (quirk B-8) and are not restated;
 * composedFrame is produced by ora_map_render_composed only. */
void ora_map_render(ora_ctx* c, const float pose_old[16], const float pose_new[16], float conf_threshold,
                    ora_frame* out) {
  const size_t Pm = (size_t)c->p.model_width * c->p.model_height;
  float inv_old[16], inv_new[16];
  om4_rigid_inverse(pose_old, inv_old);
  om4_rigid_inverse(pose_new, inv_new);
  if (c->p.compose_rendering) {
    int32_t thr = (int32_t)(c->timestamp - 100u); /* SurfelMap.cpp:873, quirk B-7 */
    o_clear_zbuf(c->zbuf_a, Pm);
    o_render_pass(c, inv_old, conf_threshold, 0, thr, c->zbuf_a, O_TIE_LOW_INDEX);
    o_render_resolve(c, inv_old, NULL, c->zbuf_a, O_TIE_LOW_INDEX, c->old_frame->vertex, c->old_frame->normal,
                     c->old_frame->semantic);
    o_clear_zbuf(c->zbuf_b, Pm);
    o_render_pass(c, inv_new, conf_threshold, 1, thr, c->zbuf_b, O_TIE_LOW_INDEX);
    o_render_resolve(c, inv_new, NULL, c->zbuf_b, O_TIE_LOW_INDEX, c->new_frame->vertex, c->new_frame->normal,
                     c->new_frame->semantic);
    /* K5 render_compose.frag:26-48 */
    const float max_distance = c->p.max_loop_closure_distance;
    for (size_t pix = 0; pix < Pm; ++pix) {
      suma_float4 v = c->new_frame->vertex[pix], n = c->new_frame->normal[pix], s = c->new_frame->semantic[pix];
      suma_float4 ov = c->old_frame->vertex[pix], on = c->old_frame->normal[pix], os = c->old_frame->semantic[pix];
      int valid = (ov.w > 0.5f && on.w > 0.5f);
      int new_valid = (v.w > 0.5f && n.w > 0.5f);
      if (!new_valid && valid &&
          (v.w < 0.5f || ov3_len(ov3_sub(ov3_make(v.x, v.y, v.z), ov3_make(ov.x, ov.y, ov.z))) < max_distance)) {
        v = ov;
        n = on;
        s = os;
      }
      out->vertex[pix] = v;
      out->normal[pix] = n;
      out->semantic[pix] = s;
    }
  } else {
    /* SurfelMap.cpp:976-1018: one pass, threshold 0, then copies */
    o_clear_zbuf(c->zbuf_a, Pm);
    o_render_pass(c, inv_old, conf_threshold, 1, 0, c->zbuf_a, O_TIE_LOW_INDEX);
    o_render_resolve(c, inv_old, NULL, c->zbuf_a, O_TIE_LOW_INDEX, out->vertex, out->normal, out->semantic);
    ora_frame_copy(c->new_frame, out);
    ora_frame_copy(c->old_frame, out);
  }
}

/* SurfelMap::render_active, SurfelMap.cpp:1023-1069.  Only COLOR0/COLOR1 are re-attached
 * (:1047-1048): the semantic map of newMapFrame_ is NOT refreshed by this call (the third colour
 * attachment still points at the frame given to the last render()).  Restated as such. */
void ora_map_render_active(ora_ctx* c, const float pose[16], float conf_threshold) {
  const size_t Pm = (size_t)c->p.model_width * c->p.model_height;
  float inv[16];
  om4_rigid_inverse(pose, inv);
  int32_t thr = (int32_t)(c->timestamp - 100u);
  o_clear_zbuf(c->zbuf_b, Pm);
  o_render_pass(c, inv, conf_threshold, 1, thr, c->zbuf_b, O_TIE_LOW_INDEX);
  o_render_resolve(c, inv, NULL, c->zbuf_b, O_TIE_LOW_INDEX, c->new_frame->vertex, c->new_frame->normal, NULL);
}

/* SurfelMap::render_inactive, SurfelMap.cpp:1071-1114 (same attachment remark) */
void ora_map_render_inactive(ora_ctx* c, const float pose[16], float conf_threshold) {
  const size_t Pm = (size_t)c->p.model_width * c->p.model_height;
  float inv[16];
  om4_rigid_inverse(pose, inv);
  int32_t thr = (int32_t)(c->timestamp - 100u);
  o_clear_zbuf(c->zbuf_a, Pm);
  o_render_pass(c, inv, conf_threshold, 0, thr, c->zbuf_a, O_TIE_LOW_INDEX);
  o_render_resolve(c, inv, NULL, c->zbuf_a, O_TIE_LOW_INDEX, c->old_frame->vertex, c->old_frame->normal, NULL);
}

/* SurfelMap::render_composed, SurfelMap.cpp:1116-1165: old pass then new pass, GL_LEQUAL, one
 * depth buffer, only vertex / normal attachments are switched to composedFrame_. */
void ora_map_render_composed(ora_ctx* c, const float pose_old[16], const float pose_new[16], float conf_threshold) {
  const size_t Pm = (size_t)c->p.model_width * c->p.model_height;
  float inv_old[16], inv_new[16];
  om4_rigid_inverse(pose_old, inv_old);
  om4_rigid_inverse(pose_new, inv_new);
  int32_t thr = (int32_t)(c->timestamp - 100u);
  o_clear_zbuf(c->zbuf_a, Pm);
  o_render_pass(c, inv_old, conf_threshold, 0, thr, c->zbuf_a, O_TIE_HIGH_INDEX_OLD);
  o_render_pass(c, inv_new, conf_threshold, 1, thr, c->zbuf_a, O_TIE_HIGH_INDEX_NEW);
  o_render_resolve(c, inv_old, inv_new, c->zbuf_a, O_TIE_HIGH_INDEX_OLD, c->composed_frame->vertex,
                   c->composed_frame->normal, NULL);
}

/* ------------------------------------------------------------------------------------------
 * K7 .. K12: SurfelMap::update
 * ---------------------------------------------------------------------------------------- */

/* K7 gen_indexmap.vert:62-81: point splat of (id + 1), GL_LESS. */
static void o_k7_indexmap(ora_ctx* c, const float* inv_pose) {
  const ora_proj q = o_proj_data(&c->p);
  const int32_t W = (int32_t)c->p.data_width, H = (int32_t)c->p.data_height;
  const size_t P = (size_t)W * H;
  o_clear_zbuf(c->zbuf_data, P);
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (uint32_t i = 0; i < c->n_surfels; ++i) {
    ov3 p, n;
    o_surfel_to_sensor(c, inv_pose, &c->surfels[i], &p, &n);
    float lp = ov3_len(p);
    if (!(ov3_dot(n, ov3_divs(ov3_neg(p), lp)) > 0.01f)) continue;
    ov3 pr = o_project01(&q, p);
    float fx = sdm_floor(pr.x * q.width), fy = sdm_floor(pr.y * q.height);
    if (!(fx >= 0.0f && fx < q.width && fy >= 0.0f && fy < q.height)) continue;
    float zn = 2.0f * pr.z - 1.0f;
    if (!(zn >= -1.0f && zn <= 1.0f)) continue;
    uint64_t key = ((uint64_t)o_depth24(0.5f * zn + 0.5f) << 32) | i;
    size_t pix = (size_t)(int32_t)fy * W + (size_t)(int32_t)fx;
    o_zmin(&c->zbuf_data[pix], key);
  }
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (size_t pix = 0; pix < P; ++pix)
    c->index_map[pix] = (c->zbuf_data[pix] == O_EMPTY) ? 0u : (uint32_t)(c->zbuf_data[pix] & 0xffffffffu) + 1u;
}

/* K8 init_radiusConf.vert:34-68 (quirk B-3: the confidence channel stays 0) */
static void o_k8_radius(ora_ctx* c, const ora_frame* f, const o_map_consts* k) {
  const size_t P = (size_t)c->p.data_width * c->p.data_height;
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (size_t pix = 0; pix < P; ++pix) {
    suma_float4 v = f->vertex[pix], n = f->normal[pix];
    ov3 vv = ov3_make(v.x, v.y, v.z), nn = ov3_make(n.x, n.y, n.z);
    float d = ov3_len(vv);
    ov3 view_dir = ov3_divs(ov3_neg(vv), d);
    float angle = ov3_dot(nn, view_dir);
    float valid = 0.0f, radius = 0.0f;
    if (v.w > 0.5f && n.w > 0.5f && angle > k->radconf_angle_thresh) {
      valid = 1.0f;
      radius = ((1.41f * d) * k->pixel_size) / of_clamp(ov3_dot(nn, ov3_divs(ov3_neg(vv), d)), 0.5f, 1.0f);
      radius = of_min(of_max(radius, c->p.min_radius), c->p.max_radius);
    }
    c->radius_conf[pix] = o_f4(radius, 0.0f, 0.0f, valid);
  }
}

/* update_surfels.vert:113-124 slerp().  Deviation (documented): when
 * sin(omega) is not > 0 (identical normals, or |dot| rounding above 1) the GLSL yields NaN and
 * poisons the surfel; here v0 is returned instead. */
static ov3 o_slerp(ov3 v0, ov3 v1, float weight) {
  float omega = sdm_acos(ov3_dot(ov3_normalize(v0), ov3_normalize(v1)));
  float so = sdm_sin(omega);
  if (!(so > 0.0f)) return v0;
  float eta = 1.0f / so;
  float w0 = eta * sdm_sin(weight * omega);
  float w1 = eta * sdm_sin((1.0f - weight) * omega);
  return ov3_add(ov3_scale(w0, v0), ov3_scale(w1, v1));
}

/* Stable in-place compaction of the flagged records: chunks compact themselves side by side (a chunk only
 * moves records towards its own start), then the chunk fronts are moved down in order. */
#define O_CHUNK 65536u
static uint32_t o_compact_inplace(ora_ctx* c, suma_surfel* buf, const uint8_t* flag, uint32_t n, uint32_t cap) {
  const uint32_t nchunks = (n + O_CHUNK - 1) / O_CHUNK;
  uint32_t* cnt = (uint32_t*)malloc((nchunks + 1) * sizeof(uint32_t));
#pragma omp parallel for num_threads(c->threads) schedule(dynamic, 1)
  for (uint32_t ch = 0; ch < nchunks; ++ch) {
    const uint32_t lo = ch * O_CHUNK, hi = (lo + O_CHUNK < n) ? lo + O_CHUNK : n;
    uint32_t w = lo;
    for (uint32_t i = lo; i < hi; ++i)
      if (flag[i]) {
        if (w != i) buf[w] = buf[i];
        w++;
      }
    cnt[ch] = w - lo;
  }
  uint32_t n_out = 0;
  for (uint32_t ch = 0; ch < nchunks; ++ch) {
    uint32_t take = cnt[ch];
    if (n_out + take > cap) take = cap - n_out; /* TF buffer full: later records are dropped */
    if (take && n_out != ch * O_CHUNK) memmove(buf + n_out, buf + (size_t)ch * O_CHUNK, (size_t)take * sizeof(suma_surfel));
    n_out += take;
  }
  free(cnt);
  return n_out;
}

/* stable compaction of the flagged records of src behind dst[0 .. n_out): counts per chunk, prefix, copies */
static uint32_t o_compact_copy(ora_ctx* c, suma_surfel* dst, uint32_t n_out, const suma_surfel* src,
                               const uint8_t* flag, uint32_t n, uint32_t cap) {
  const uint32_t nchunks = (n + O_CHUNK - 1) / O_CHUNK;
  uint32_t* off = (uint32_t*)malloc((nchunks + 1) * sizeof(uint32_t));
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (uint32_t ch = 0; ch < nchunks; ++ch) {
    const uint32_t lo = ch * O_CHUNK, hi = (lo + O_CHUNK < n) ? lo + O_CHUNK : n;
    uint32_t k = 0;
    for (uint32_t i = lo; i < hi; ++i) k += flag[i];
    off[ch + 1] = k;
  }
  off[0] = n_out;
  for (uint32_t ch = 0; ch < nchunks; ++ch) off[ch + 1] += off[ch];
#pragma omp parallel for num_threads(c->threads) schedule(dynamic, 1)
  for (uint32_t ch = 0; ch < nchunks; ++ch) {
    const uint32_t lo = ch * O_CHUNK, hi = (lo + O_CHUNK < n) ? lo + O_CHUNK : n;
    uint32_t w = off[ch];
    for (uint32_t i = lo; i < hi; ++i)
      if (flag[i]) {
        if (w < cap) dst[w] = src[i];
        w++;
      }
  }
  uint32_t total = off[nchunks];
  free(off);
  return total < cap ? total : cap;
}

static uint8_t* o_scratch_flags(ora_ctx* c, size_t n) {
  if (c->scratch_flags_cap < n) {
    free(c->scratch_flags);
    c->scratch_flags_cap = n + n / 4 + 1024;
    c->scratch_flags = (uint8_t*)malloc(c->scratch_flags_cap);
  }
  return c->scratch_flags;
}

/* K9 update_surfels.vert:140-334 + update_surfels.geom:30-43 (emit iff valid) +
 * update_surfels.frag:9-12 (integration mask), output stable in input order (transform feedback). */
static void o_k9_update(ora_ctx* c, const float* pose, const float* inv_pose, const ora_frame* f,
                        const o_map_consts* k) {
  const suma_params* p = &c->p;
  const ora_proj q = o_proj_data(p);
  const int32_t W = (int32_t)p->data_width, H = (int32_t)p->data_height;
  const int32_t timestamp = (int32_t)c->timestamp;
  const float upper_stability_bound = 20.0f;
  memset(c->integrated, 0, (size_t)W * H);
  /* phase 1 (any number of threads): surfel i -> its updated record at updated[i] + keep flag;
   * phase 2: stable compaction in input order, as the transform feedback delivers it */
  uint8_t* keep_flag = o_scratch_flags(c, c->n_surfels);
#pragma omp parallel for num_threads(c->threads) schedule(dynamic, 2048)
  for (uint32_t i = 0; i < c->n_surfels; ++i) {
    const suma_surfel in = c->surfels[i];
    int32_t surfel_age = timestamp - (int32_t)in.timestamp;
    int32_t creation_timestamp = (int32_t)in.count;
    const float* Ps = c->poses + 16 * (size_t)creation_timestamp;
    ov3 old_position = om4_point(Ps, ov3_make(in.x, in.y, in.z));
    ov3 old_normal = om4_dir(Ps, ov3_make(in.nx, in.ny, in.nz));
    float old_radius = in.radius, old_confidence = in.confidence, old_weight = in.weight;

    int keep = 1;
    if (old_confidence < p->confidence_threshold && p->use_stability) keep = (surfel_age < p->unstable_age);
    suma_surfel out = in;
    out.color = o_pack(0.3f, 0.3f, 0.3f);

    ov3 vertex = om4_point(inv_pose, old_position);
    ov3 normal = ov3_normalize(om4_dir(inv_pose, old_normal));
    int visible = ov3_dot(normal, ov3_divs(ov3_neg(vertex), ov3_len(vertex))) > 0.0f;
    ov3 pr = o_project01(&q, vertex);
    float imx = sdm_floor(pr.x * q.width) + 0.5f, imy = sdm_floor(pr.y * q.height) + 0.5f, imz = pr.z;
    /* texel fetch at (imx, imy): NEAREST/exact centre, border (0) outside or for NaN */
    int in_tex = (imx >= 0.0f && imx < q.width && imy >= 0.0f && imy < q.height);
    int32_t tx = in_tex ? (int32_t)sdm_floor(imx) : -1, ty = in_tex ? (int32_t)sdm_floor(imy) : -1;
    suma_float4 dv = o_texel(f->vertex, W, H, tx, ty), dn = o_texel(f->normal, W, H, tx, ty);
    int valid = (dv.w > 0.5f) && (dn.w > 0.5f);
    /* quirk B-6: all(lessThan(img, dim)) && !all(lessThan(img, 0)) */
    int inside = (imx < q.width && imy < q.height && imz < 1.0f) && !(imx < 0.0f && imy < 0.0f && imz < 0.0f);

    float penalty = 0.0f;
    float update_confidence = k->log_prior;
    int mark_pixel = 0;

    if (valid && inside && visible) {
      suma_float4 ds = o_texel(f->semantic, W, H, tx, ty);
      float data_label = ds.x * 255.0f, data_prob = ds.w;
      float model_label = in.r * 255.0f, model_prob = in.w;
      if (sdm_round(data_label) != sdm_round(model_label)) {
        if (o_is_dynamic_label(model_label)) penalty = 1.0f;
      }
      const float* Ps_inv = c->poses_inv + 16 * (size_t)creation_timestamp;
      ov3 v = ov3_make(dv.x, dv.y, dv.z), n = ov3_make(dn.x, dn.y, dn.z);
      ov3 v_global = om4_point(pose, v);
      ov3 n_global = ov3_normalize(om4_dir(pose, n));
      ov3 view_dir = ov3_divs(ov3_neg(v), ov3_len(v));
      float distance = fabsf(ov3_dot(old_normal, ov3_sub(v_global, old_position)));
      float angle = ov3_len(ov3_cross(n_global, old_normal));
      suma_float4 rc = o_texel(c->radius_conf, W, H, tx, ty);
      float new_radius = rc.x, new_confidence = rc.y;

      if ((distance < p->map_max_distance) && (angle < k->update_angle_thresh)) {
        mark_pixel = 1; /* gl_Position inside the viewport: update_surfels.vert:219 */
        float confidence = old_confidence + new_confidence;
        out.confidence = confidence;
        out.timestamp = (uint32_t)timestamp;
        float avg_radius = of_min(new_radius, old_radius);
        avg_radius = of_max(avg_radius, 0.0f); /* update program's min_radius uniform is 0, SurfelMap.cpp:422 */
        out.radius = avg_radius;
        keep = 1;
        out.color = o_pack(0.0f, 0.7f, 0.0f);
        out.count = (float)creation_timestamp;

        float a = angle, d = distance;
        float pst = p->p_stable;
        if (p->confidence_mode == 1 || p->confidence_mode == 3)
          pst *= sdm_exp((-a * a) / (p->sigma_angle * p->sigma_angle));
        if (p->confidence_mode == 2 || p->confidence_mode == 3)
          pst *= sdm_exp((-d * d) / (p->sigma_distance * p->sigma_distance));
        pst = of_clamp(pst, k->p_unstable, 1.0f);
        update_confidence = sdm_log(pst / (1.0f - pst));

        if ((new_radius < old_radius && timestamp - creation_timestamp < p->active_timestamps) || p->update_always) {
          float w1 = 0.9f, w2 = 0.1f;
          if (p->weighting_scheme > 0) {
            w1 = old_weight;
            w2 = 1.0f;
            if (p->weighting_scheme == 2) w2 = ov3_dot(n, view_dir);
            out.weight = of_min(p->max_weight, w1 + w2);
            float sum = w1 + w2;
            w1 /= sum;
            w2 /= sum;
          }
          ov3 avg_position = ov3_add(ov3_scale(w1, old_position), ov3_scale(w2, v_global));
          ov3 avg_normal = o_slerp(old_normal, n_global, w1);
          float avg_prob;
          if (sdm_round(data_label) != sdm_round(model_label))
            avg_prob = w1 * model_prob + w2 * (1.0f - data_prob);
          else
            avg_prob = w1 * model_prob + w2 * data_prob;
          out.w = avg_prob;
          if (p->averaging_scheme == 1) {
            avg_position = ov3_add(old_position, ov3_scale(w2 * distance, old_normal));
            avg_normal = o_slerp(old_normal, n_global, w1);
          }
          avg_normal = ov3_normalize(avg_normal);
          avg_position = om4_point(Ps_inv, avg_position);
          avg_normal = om4_dir(Ps_inv, avg_normal);
          out.x = avg_position.x;
          out.y = avg_position.y;
          out.z = avg_position.z;
          out.radius = avg_radius;
          out.nx = avg_normal.x;
          out.ny = avg_normal.y;
          out.nz = avg_normal.z;
          out.confidence = confidence;
          out.color = o_pack(1.0f, 0.0f, 1.0f);
        }
      } else {
        int32_t idx = (int32_t)c->index_map[(size_t)ty * W + tx] - 1;
        if (idx == (int32_t)i) {
          update_confidence = sdm_log(k->p_unstable / (1.0f - k->p_unstable));
          out.color = o_pack(0.0f, 1.0f, 1.0f);
        }
      }
    }
    update_confidence = update_confidence - penalty;
    if (p->use_stability)
      out.confidence = of_min((old_confidence + update_confidence) - k->log_prior, upper_stability_bound);
    else
      out.confidence = old_confidence;
    if (out.confidence < k->log_unstable && p->use_stability) keep = 0;

    if (keep) {
      /* rasterised point of the surviving surfel marks the measurement as integrated; clipped
       * unless z_ndc = 2*z01 - 1 lies in [-1, 1] (x, y are texel centres inside the viewport) */
      if (mark_pixel) {
        float zn = 2.0f * imz - 1.0f;
        if (zn >= -1.0f && zn <= 1.0f) __atomic_store_n(&c->integrated[(size_t)ty * W + tx], (uint8_t)1, __ATOMIC_RELAXED);
      }
      c->updated[i] = out;
    }
    keep_flag[i] = (uint8_t)(keep != 0);
  }
  c->n_updated = o_compact_inplace(c, c->updated, keep_flag, c->n_surfels, p->max_surfels);
}

/* K10 gen_surfels.vert:38-52 + gen_surfels.geom:109-145; emission order = vbo_img_coords_
 * (x-major, SurfelMap.cpp:88-92) */
static void o_k10_generate(ora_ctx* c, const ora_frame* f, const o_map_consts* k) {
  const int32_t W = (int32_t)c->p.data_width, H = (int32_t)c->p.data_height;
  const uint32_t cap = 2u * (uint32_t)W * (uint32_t)H;
  const float color = o_pack(0.0f, 0.0f, 1.0f);
  uint32_t n_out = 0;
  for (int32_t x = 0; x < W; ++x) {
    for (int32_t y = 0; y < H; ++y) {
      size_t pix = (size_t)y * W + x;
      suma_float4 v = f->vertex[pix], n = f->normal[pix];
      int invalid = (v.w < 1.0f) || (n.w < 1.0f);
      invalid = invalid || (c->radius_conf[pix].w < 0.5f);
      int integrated = c->integrated[pix] != 0;
      ov3 vv = ov3_make(v.x, v.y, v.z), nn = ov3_make(n.x, n.y, n.z);
      ov3 view_dir = ov3_divs(ov3_neg(vv), ov3_len(vv));
      if (!(!invalid && !integrated && (ov3_dot(nn, view_dir) > 0.01f))) continue;
      ov3 ng = ov3_normalize(nn);
      suma_surfel s;
      s.x = v.x;
      s.y = v.y;
      s.z = v.z;
      s.radius = c->radius_conf[pix].x;
      s.nx = ng.x;
      s.ny = ng.y;
      s.nz = ng.z;
      s.confidence = k->log_prior;
      s.timestamp = c->timestamp;
      s.color = color;
      s.weight = 1.0f;
      s.count = (float)(int32_t)c->timestamp;
      suma_float4 sem = f->semantic[pix];
      s.r = sem.x;
      s.g = sem.y;
      s.b = sem.z;
      s.w = sem.w;
      float semantic_label = sem.x * 255.0f;
      if (o_is_dynamic_label(semantic_label)) s.confidence = k->log_prior - 0.5f;
      if (n_out < cap) c->data_surfels[n_out++] = s;
    }
  }
  c->n_data = n_out;
}

static inline void o_submap_center(const ora_ctx* c, int32_t i, int32_t j, float* cx, float* cy) {
  *cx = (float)(2.0 * i * c->p.submap_extent); /* SurfelMap.cpp:704-706 */
  *cy = (float)(2.0 * j * c->p.submap_extent);
}

/* K11 copy_surfels.vert:38-56, SurfelMap.cpp:667-698 */
static void o_k11_copy(ora_ctx* c) {
  float cx, cy;
  o_submap_center(c, c->origin_i, c->origin_j, &cx, &cy);
  float extent = 2.0f * (float)c->p.submap_dimension * c->p.submap_extent + c->p.submap_extent;
  if (c->p.partial_extraction && c->n_extraction > 0) extent += 2.0f * c->p.submap_extent;
  uint32_t n_out = 0;
  for (int src = 0; src < 2; ++src) {
    const suma_surfel* buf = src == 0 ? c->updated : c->data_surfels;
    uint32_t n = src == 0 ? c->n_updated : c->n_data;
    uint8_t* sel = o_scratch_flags(c, n);
#pragma omp parallel for num_threads(c->threads) schedule(static)
    for (uint32_t i = 0; i < n; ++i) {
      const suma_surfel* s = &buf[i];
      ov3 pos = om4_point(c->poses + 16 * (size_t)(int32_t)s->count, ov3_make(s->x, s->y, s->z));
      sel[i] = !((int32_t)s->timestamp < 0 || fabsf(pos.x - cx) > extent || fabsf(pos.y - cy) > extent);
    }
    n_out = o_compact_copy(c, c->surfels, n_out, buf, sel, n, c->p.max_surfels);
  }
  c->n_surfels = n_out;
}

static ora_submap_cache* o_cache_get(ora_ctx* c, int32_t i, int32_t j) {
  for (uint32_t k = 0; k < c->n_caches; ++k)
    if (c->caches[k].i == i && c->caches[k].j == j) return &c->caches[k];
  if (c->n_caches == c->cap_caches) {
    c->cap_caches = c->cap_caches ? 2 * c->cap_caches : 64;
    c->caches = (ora_submap_cache*)realloc(c->caches, c->cap_caches * sizeof(ora_submap_cache));
  }
  ora_submap_cache* e = &c->caches[c->n_caches++];
  e->i = i;
  e->j = j;
  e->surfels = NULL;
  e->n = 0;
  return e;
}
static void o_extraction_push(ora_ctx* c, int32_t i, int32_t j) {
  if (c->n_extraction == c->cap_extraction) {
    c->cap_extraction = c->cap_extraction ? 2 * c->cap_extraction : 64;
    c->extraction = (int32_t*)realloc(c->extraction, 2 * c->cap_extraction * sizeof(int32_t));
  }
  c->extraction[2 * c->n_extraction] = i;
  c->extraction[2 * c->n_extraction + 1] = j;
  c->n_extraction++;
}
static void o_append_cached(ora_ctx* c, int32_t i, int32_t j) {
  ora_submap_cache* e = o_cache_get(c, i, j);
  for (uint32_t k = 0; k < e->n; ++k)
    if (c->n_surfels < c->p.max_surfels) c->surfels[c->n_surfels++] = e->surfels[k];
}

/* K12 extract_surfels.vert:46-64, SurfelMap::extractSurfels SurfelMap.cpp:708-742 */
static void o_extract(ora_ctx* c, int partially) {
  while (c->n_extraction > 0) {
    c->n_extraction--;
    int32_t i = c->extraction[2 * c->n_extraction], j = c->extraction[2 * c->n_extraction + 1];
    float cx, cy;
    o_submap_center(c, i, j, &cx, &cy);
    ora_submap_cache* e = o_cache_get(c, i, j);
    free(e->surfels);
    e->surfels = (suma_surfel*)malloc((size_t)O_EXTRACT_CAPACITY * sizeof(suma_surfel));
    uint32_t n = 0;
    for (uint32_t k = 0; k < c->n_surfels; ++k) {
      const suma_surfel* s = &c->surfels[k];
      ov3 pos = om4_point(c->poses + 16 * (size_t)(int32_t)s->count, ov3_make(s->x, s->y, s->z));
      if (fabsf(pos.x - cx) > c->p.submap_extent || fabsf(pos.y - cy) > c->p.submap_extent) continue;
      if (n < O_EXTRACT_CAPACITY) e->surfels[n++] = *s;
    }
    e->n = n;
    e->surfels = (suma_surfel*)realloc(e->surfels, (size_t)(n ? n : 1) * sizeof(suma_surfel));
    c->last_extract_i = i;
    c->last_extract_j = j;
    c->n_extractions_done++;
    if (partially) break;
  }
}

/* SurfelMap::updateActiveSubmaps, SurfelMap.cpp:744-824 */
static void o_update_active_submaps(ora_ctx* c, const float* pose) {
  const int32_t dim = c->p.submap_dimension;
  const float ext = c->p.submap_extent;
  float cx, cy;
  o_submap_center(c, c->origin_i, c->origin_j, &cx, &cy);
  float changex = pose[12] - cx, changey = pose[13] - cy;
  const float factor = 1.1f;
  if (fabsf(changex) > factor * ext || fabsf(changey) > factor * ext) {
    if (fabsf(changex) > factor * ext) {
      int32_t dir = (changex < 0) ? -1 : 1;
      for (int32_t k = -dim; k <= dim; ++k) o_extraction_push(c, c->origin_i - dir * dim, c->origin_j + k);
      c->origin_i += dir;
      for (int32_t k = -dim; k <= dim; ++k) o_append_cached(c, c->origin_i + dir * dim, c->origin_j + k);
    }
    if (fabsf(changey) > factor * ext) {
      int32_t dir = (changey < 0) ? -1 : 1;
      for (int32_t r = -dim; r <= dim; ++r) o_extraction_push(c, c->origin_i + r, c->origin_j - dir * dim);
      c->origin_j += dir;
      for (int32_t r = -dim; r <= dim; ++r) o_append_cached(c, c->origin_i + r, c->origin_j + dir * dim);
    }
  }
  if (c->n_extraction > 0) o_extract(c, c->p.partial_extraction);
}

/* SurfelMap::update, SurfelMap.cpp:492-584 */
void ora_map_update(ora_ctx* c, const float pose[16], const ora_frame* frame) {
  const o_map_consts k = o_consts(&c->p);
  if (c->timestamp < c->p.max_poses) o_set_pose(c, c->timestamp, pose);
  float inv_pose[16];
  om4_rigid_inverse(pose, inv_pose);
  o_k7_indexmap(c, inv_pose);
  o_k8_radius(c, frame, &k);
  o_k9_update(c, pose, inv_pose, frame, &k);
  o_k10_generate(c, frame, &k);
  o_k11_copy(c);
  o_update_active_submaps(c, pose);
  c->timestamp += 1;
}
