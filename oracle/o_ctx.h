/* oracle/o_ctx.h -- TEST INFRASTRUCTURE. Internal state of the CPU oracle. */
#ifndef ORACLE_O_CTX_H_
#define ORACLE_O_CTX_H_

#include <stdlib.h>

#include "o_math.h"
#include "suma_oracle.h"

typedef struct ora_submap_cache {
  int32_t i, j;
  suma_surfel* surfels;
  uint32_t n;
} ora_submap_cache;

struct ora_ctx {
  suma_params p;
  int threads;
  uint8_t* scratch_flags; /* per-item selection flags of the compacting passes */
  size_t scratch_flags_cap;

  /* --- surfel map state (SurfelMap.h:83-208) --- */
  uint32_t timestamp; /* SurfelMap::timestamp_ */
  suma_surfel* surfels;
  uint32_t n_surfels;
  suma_surfel* updated;
  uint32_t n_updated;
  suma_surfel* data_surfels;
  uint32_t n_data;
  float* poses;     /* max_poses x 16, column-major */
  float* poses_inv; /* rigid inverses, same layout */
  ora_frame *old_frame, *new_frame, *composed_frame;
  uint32_t* index_map;
  suma_float4* radius_conf;
  uint8_t* integrated;
  uint64_t* zbuf_a; /* model-sized z-buffers */
  uint64_t* zbuf_b;
  uint64_t* zbuf_data; /* data-sized */
  /* submaps (SurfelMap.cpp:744-824) */
  int32_t origin_i, origin_j;
  ora_submap_cache* caches;
  uint32_t n_caches, cap_caches;
  int32_t* extraction; /* pending (i,j) pairs, used as a stack */
  uint32_t n_extraction, cap_extraction;
  /* record of the last K12 extraction, for stage-by-stage tests */
  int32_t last_extract_i, last_extract_j;
  uint32_t n_extractions_done;
};

/* derived constants, computed the way the reference's setParameters() compute them */
typedef struct ora_proj {
  float fov_up, fov_down, fov, min_depth, max_depth, width, height;
} ora_proj;

static inline ora_proj o_proj_data(const suma_params* p) {
  ora_proj q;
  q.fov_up = fabsf(p->data_fov_up);
  q.fov_down = fabsf(p->data_fov_down);
  q.fov = fabsf(q.fov_up) + fabsf(q.fov_down);
  q.min_depth = p->min_depth;
  q.max_depth = p->max_depth;
  q.width = (float)p->data_width;
  q.height = (float)p->data_height;
  return q;
}
static inline ora_proj o_proj_model(const suma_params* p) {
  ora_proj q;
  q.fov_up = fabsf(p->model_fov_up);
  q.fov_down = fabsf(p->model_fov_down);
  q.fov = fabsf(q.fov_up) + fabsf(q.fov_down);
  q.min_depth = p->model_min_depth;
  q.max_depth = p->model_max_depth;
  q.width = (float)p->model_width;
  q.height = (float)p->model_height;
  return q;
}

/* The spherical projection repeated in five shaders (gen_indexmap.vert:37-52 et al.):
 * returns continuous (x01, y01, z01). */
static inline ov3 o_project01(const ora_proj* q, ov3 p) {
  float depth = ov3_len(p);
  float yaw = sdm_atan2(p.y, p.x);
  float pitch = -sdm_asin(p.z / depth);
  ov3 r;
  r.x = 0.5f * ((-yaw * SUMA_INV_PI_F) + 1.0f);
  r.y = 1.0f - ((pitch * SUMA_RAD2DEG_F) + q->fov_up) / q->fov;
  r.z = (depth - q->min_depth) / (q->max_depth - q->min_depth);
  return r;
}

static inline suma_float4 o_f4(float x, float y, float z, float w) {
  suma_float4 r = {x, y, z, w};
  return r;
}
/* NEAREST + CLAMP_TO_BORDER fetch at integer texel (border colour 0) */
static inline suma_float4 o_texel(const suma_float4* map, int32_t w, int32_t h, int32_t x, int32_t y) {
  if (x < 0 || y < 0 || x >= w || y >= h) return o_f4(0.f, 0.f, 0.f, 0.f);
  return map[(size_t)y * (size_t)w + (size_t)x];
}

void o_map_alloc(ora_ctx* c);
void o_map_free(ora_ctx* c);

/* depth-tested write of a 64-bit z-buffer key (smaller wins).  A min is order independent, so the point /
 * triangle splats may run on several threads (ora_set_threads) and still give the single-thread image. */
static inline void o_zmin(uint64_t* p, uint64_t key) {
  uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (key < cur && !__atomic_compare_exchange_n(p, &cur, key, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
}

#endif
