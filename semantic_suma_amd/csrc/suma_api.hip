/*
 * suma_api.hip -- the C-ABI of include/suma_hip.h: context / frame management, the host side of
 * Preprocessing, Frame2Model + LieGaussNewton and SurfelMap, and the per-scan sequencing of
 * SurfelMapping::processScan (reference src/core/SurfelMapping.cpp:175-210, 323-358, 372-476,
 * 797-804).  Host logic only; all arithmetic on scan data runs in the kernels of k_*.hip.
 */
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <set>

#include <time.h>

#include "suma_internal.h"

/* contexts that are alive: a frame may outlive its context (callers destroy in any order), so suma_frame_destroy asks
 * here before it touches the context's frame-pointer caches */
static std::mutex g_ctx_mu;
static std::set<const suma_ctx*> g_live_ctx;
static bool ctx_alive(const suma_ctx* c) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  return g_live_ctx.count(c) != 0;
}

static thread_local std::string g_create_error;

extern "C" const char* suma_version(void) { return "suma-hip 0.1 (gfx950)"; }
extern "C" const char* suma_last_error(const suma_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

/* ---------------------------------------------------------------------------------------------
 * helpers
 * ------------------------------------------------------------------------------------------- */
void rigid_inverse_f(const float* m, float* out) {
  double R[9], t[3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) R[3 * c + r] = (double)m[4 * c + r];
  for (int r = 0; r < 3; ++r) t[r] = (double)m[12 + r];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) out[4 * c + r] = (float)R[3 * r + c];
  for (int r = 0; r < 3; ++r) {
    double s = (R[3 * r + 0] * t[0] + R[3 * r + 1] * t[1]) + R[3 * r + 2] * t[2];
    out[12 + r] = (float)(-s);
  }
  out[3] = out[7] = out[11] = 0.0f;
  out[15] = 1.0f;
}

static float deg2rad_f(float deg) { return (float)((double)deg * M_PI / 180.0); }

/* derived constants exactly as the reference's setParameters() compute them
 * (Preprocessing.cpp:77-100, SurfelMap.cpp:336-457) */
static void derive(suma_ctx* c) {
  const suma_params& p = c->p;
  c->pd.fov_up = fabsf(p.data_fov_up);
  c->pd.fov = fabsf(fabsf(p.data_fov_up)) + fabsf(fabsf(p.data_fov_down));
  c->pd.min_depth = p.min_depth;
  c->pd.max_depth = p.max_depth;
  c->pd.width = (float)p.data_width;
  c->pd.height = (float)p.data_height;
  c->pd.W = (int32_t)p.data_width;
  c->pd.H = (int32_t)p.data_height;
  c->pm.fov_up = fabsf(p.model_fov_up);
  c->pm.fov = fabsf(fabsf(p.model_fov_up)) + fabsf(fabsf(p.model_fov_down));
  c->pm.min_depth = p.model_min_depth;
  c->pm.max_depth = p.model_max_depth;
  c->pm.width = (float)p.model_width;
  c->pm.height = (float)p.model_height;
  c->pm.W = (int32_t)p.model_width;
  c->pm.H = (int32_t)p.model_height;
  float vfov = fabsf(p.data_fov_up) + fabsf(p.data_fov_down);
  float hfov = 360.0f;
  /* SurfelMap.cpp:342-343: 0.5f * Math::deg2rad(vfov) / uint32_t(height) -- rv::Math::deg2rad is double -> the
   * whole argument is evaluated in double (Math.h:44-47), std::tan(double), then rounded to float */
  float vpix = (float)tan(0.5 * ((double)vfov * M_PI / 180.0) / (double)p.data_height);
  float hpix = (float)tan(0.5 * ((double)hfov * M_PI / 180.0) / (double)p.data_width);
  c->mc.pixel_size = vpix < hpix ? hpix : vpix;
  c->mc.p_unstable = 1.0f - p.p_stable;
  c->mc.log_prior = (float)log((double)p.p_prior / (1.0 - (double)p.p_prior));
  c->mc.log_unstable = (float)log((double)c->mc.p_unstable / (1.0 - (double)c->mc.p_unstable));
  c->mc.radconf_angle_thresh = (float)cos((double)deg2rad_f(p.max_angle));
  /* SurfelMap.cpp:407: std::sin(Radians(float)) -- rv/geometry.h:81-84 Radians() is ((float)M_PI / 180.f) * deg in
   * float, and std::sin(float) is sinf */
  c->mc.update_angle_thresh = sinf(((float)M_PI / 180.f) * p.map_max_angle);
}

#define CK(expr)                                                                 \
  do {                                                                           \
    hipError_t e__ = (expr);                                                     \
    if (e__ != hipSuccess) {                                                     \
      c->err = std::string(#expr) + ": " + hipGetErrorString(e__);               \
      return SUMA_ERR_HIP;                                                       \
    }                                                                            \
  } while (0)

/* ctx-stream access bookkeeping of a frame (suma_frame.last_access) */
static inline void accessed(suma_ctx* c, const suma_frame* f) {
  if (f) const_cast<suma_frame*>(f)->last_access = ++c->enq_seq;
}
/* the host has just observed the completion of everything the ctx stream held */
static inline void host_synced(suma_ctx* c) { c->done_seq = c->enq_seq; }

static int fail(suma_ctx* c, int code, const char* msg) {
  c->err = msg;
  return code;
}

/* ---------------------------------------------------------------------------------------------
 * profiling
 * ------------------------------------------------------------------------------------------- */
int prof_begin(suma_ctx* c, const char* name, double bytes, uint32_t launches) {
  int id = -1;
  for (size_t i = 0; i < c->prof_names.size(); ++i)
    if (c->prof_names[i] == name) id = (int)i;
  if (id < 0) {
    id = (int)c->prof_names.size();
    c->prof_names.push_back(name);
    c->prof_ms.push_back(0.0);
    c->prof_bytes.push_back(0.0);
    c->prof_launches.push_back(0);
  }
  ProfEvent ev;
  ev.id = id;
  ev.bytes = bytes;
  ev.launches = launches;
  for (hipEvent_t* e : {&ev.a, &ev.b}) { /* events are pooled: creation is not on the per-launch path */
    if (!c->prof_pool.empty()) {
      *e = c->prof_pool.back();
      c->prof_pool.pop_back();
    } else if (hipEventCreate(e) != hipSuccess) {
      return -1;
    }
  }
  ev.stream = c->ls;
  hipEventRecord(ev.a, ev.stream);
  c->prof_events.push_back(ev);
  return (int)c->prof_events.size() - 1;
}
void prof_end(suma_ctx* c, int token) { hipEventRecord(c->prof_events[token].b, c->prof_events[token].stream); }

static void prof_collect(suma_ctx* c) {
  if (c->prof_events.empty()) return;
  hipStreamSynchronize(c->stream);
  if (c->side_stream) hipStreamSynchronize(c->side_stream);
  for (auto& ev : c->prof_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
      c->prof_ms[ev.id] += ms;
      c->prof_bytes[ev.id] += ev.bytes;
      c->prof_launches[ev.id] += ev.launches;
    }
    c->prof_pool.push_back(ev.a);
    c->prof_pool.push_back(ev.b);
  }
  c->prof_events.clear();
}

extern "C" int suma_profile_enable(suma_ctx* c, int on) {
  if (!c) return SUMA_ERR_INVALID;
  prof_collect(c);
  c->profiling = on < 0 ? 0 : (on > 3 ? 1 : on);
  if (c->prof_filter.empty()) c->prof_filter = "k6_icp_step";
  return SUMA_OK;
}
extern "C" int suma_profile_reset(suma_ctx* c) {
  if (!c) return SUMA_ERR_INVALID;
  prof_collect(c);
  for (size_t i = 0; i < c->prof_ms.size(); ++i) {
    c->prof_ms[i] = 0.0;
    c->prof_bytes[i] = 0.0;
    c->prof_launches[i] = 0;
  }
  return SUMA_OK;
}
extern "C" int suma_profile_get(suma_ctx* c, suma_kernel_time* out, uint32_t cap) {
  if (!c) return SUMA_ERR_INVALID;
  prof_collect(c);
  uint32_t n = (uint32_t)c->prof_names.size();
  for (uint32_t i = 0; i < n && i < cap; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    snprintf(out[i].name, sizeof(out[i].name), "%s", c->prof_names[i].c_str());
    out[i].launches = c->prof_launches[i];
    out[i].total_ms = c->prof_ms[i];
    out[i].bytes = c->prof_bytes[i];
  }
  return (int)n;
}

/* ---------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------- */
static int frame_create_raw(suma_ctx* c, uint32_t w, uint32_t h, suma_frame** out) {
  suma_frame* f = new (std::nothrow) suma_frame();
  if (!f) return fail(c, SUMA_ERR_NOMEM, "out of host memory");
  f->ctx = c;
  f->width = w;
  f->height = h;
  size_t P = (size_t)w * h;
  float4* base = nullptr;
  hipError_t e = hipMalloc((void**)&base, 3 * P * sizeof(float4));
  if (e != hipSuccess) {
    delete f;
    c->err = std::string("hipMalloc(frame): ") + hipGetErrorString(e);
    return SUMA_ERR_HIP;
  }
  hipMemsetAsync(base, 0, 3 * P * sizeof(float4), c->stream);
  for (int m = 0; m < 3; ++m) f->map[m] = base + m * P;
  /* the memset is ctx-stream work on this frame: a side-stream preprocessing into it must be ordered behind it when
   * the ctx stream has a backlog (round-4 advisor: last_access stayed 0, so the memset could land on fresh maps) */
  accessed(c, f);
  *out = f;
  return SUMA_OK;
}

static int read_state(suma_ctx* c) {
  CK(hipMemcpyAsync(c->h_ds, c->ds, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  c->known_surfels = c->h_ds->n_surfels;
  return SUMA_OK;
}

static int map_reset_impl(suma_ctx* c) {
  c->cache_bound = 0;
  if (c->k7.valid) CK(launch_clear_index_zbuf(c)); /* an index-map splat nobody will consume */
  CK(hipMemsetAsync(c->ds, 0, sizeof(DevState), c->stream));
  /* slot ids restart at 0 with the cleared index: the device table must not keep the previous sequence's
   * (offset, count) pairs (the compaction would count them as live blocks) */
  if (c->cache_slots) CK(hipMemsetAsync(c->cache_slots, 0, (size_t)c->cache_slots_cap * sizeof(CacheSlot), c->stream));
  c->cache_compactions = 0;
  c->cache_nothing_stale = false;
  c->flagged.valid = false;
  c->k7_spec.on = true;
  c->k7_spec.have_last = false;
  CK(launch_fill_identity_poses(c));
  c->timestamp = 0;
  c->cur = 0;
  c->origin_i = c->origin_j = 0;
  c->cache_index.clear();
  c->extraction.clear();
  c->known_surfels = 0;
  c->map_version++;
  c->rendered.valid = false;
  c->k7.valid = false;
  c->k8_fused_frame = nullptr; /* the timestamp restarts at 0: a stale (frame, stamp) pair must not match again */
  return SUMA_OK;
}

/* the optional vertex-map filters (suma_types.h): what the reference would throw on, and what k_filters.hip covers */
static const char* filter_params_error(const suma_params* p) {
  if (p->filter_sampling != SUMA_FILTER_SAMPLING_GL_INITIAL && p->filter_sampling != SUMA_FILTER_SAMPLING_NEAREST)
    return "filter_sampling: unknown value";
  if (p->filter_vertexmap && !(p->bilateral_sigma_space > 0.0f && p->bilateral_sigma_range > 0.0f))
    return "filter_vertexmap needs bilateral_sigma_space and bilateral_sigma_range > 0 (config/default.xml holds no "
           "bilateral_sigma_space; Preprocessing.cpp:86 would throw on the missing key)";
  /* k_render.hip keeps window coordinates in 1/256 pixel with |X| < 2^21 (x01 in [-0.5, 1.5] for a quad across the
   * seam): 1.5 * 256 * W < 2^21; its edge functions and exact barycentric divisions are proven on that range */
  if (p->model_width > SUMA_MAX_MODEL_WIDTH)
    return "model_width above 5461 columns is outside the range of the rasteriser's fixed-point window coordinates "
           "(k_render.hip: |X| < 2^21 in 1/256 pixel)";
  if ((p->filter_vertexmap || p->avg_vertexmap) && (p->data_width > 8192 || p->data_height > 8192))
    return "avg_vertexmap / filter_vertexmap: images above 8192 texels per side are not covered (k_filters.hip)";
  return nullptr;
}

extern "C" int suma_ctx_create(const suma_params* params, int hip_device, suma_ctx** out) {
  if (!params || !out) {
    g_create_error = "suma_ctx_create: null argument";
    return SUMA_ERR_INVALID;
  }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) +
                     "); this library has no CPU fallback";
    return SUMA_ERR_HIP;
  }
  if (hip_device < 0 || hip_device >= ndev) {
    g_create_error = "suma_ctx_create: device index out of range";
    return SUMA_ERR_INVALID;
  }
  if (params->data_width == 0 || params->data_height == 0 || params->model_width == 0 || params->model_height == 0 ||
      params->max_surfels == 0 || params->max_poses == 0) {
    g_create_error = "suma_ctx_create: zero image size or capacity";
    return SUMA_ERR_INVALID;
  }
  if (const char* fe = filter_params_error(params)) {
    g_create_error = std::string("suma_ctx_create: ") + fe;
    return SUMA_ERR_INVALID;
  }
  suma_ctx* c = new (std::nothrow) suma_ctx();
  if (!c) {
    g_create_error = "out of host memory";
    return SUMA_ERR_NOMEM;
  }
  c->p = *params;
  c->device = hip_device;
  memset(&c->het, 0, sizeof(c->het));
  c->icp_iteration0 = 0;
  c->profiling = 0;
  c->prof_tick = 0;
  c->epoch = 0;
  c->scan_cap = 0;
  c->scan_points = nullptr;
  c->scan_labels = c->scan_probs = nullptr;
  c->icp_current = c->icp_model = nullptr;
  c->obj_set = false;
  c->side_stream = nullptr;
  c->sync_flags = nullptr;
  c->zbuf_k1 = nullptr;
  c->filt_temp = nullptr;
  c->filt_sort = nullptr;
  c->filt_sort_tmp = nullptr;
  c->filt_sort_tmp_bytes = 0;
  c->filt_cap = 0;
  derive(c);
  c->P = (size_t)params->data_width * params->data_height;
  c->Pm = (size_t)params->model_width * params->model_height;
  int rc = SUMA_OK;
  auto body = [&]() -> int {
    CK(hipSetDevice(hip_device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->ls = c->stream;
    CK(hipMalloc((void**)&c->sync_flags, 16 * sizeof(uint32_t)));
    CK(hipMemsetAsync(c->sync_flags, 0, 16 * sizeof(uint32_t), c->stream));
    c->pre_seq = 0;
    c->gate_pending = 0;
    const size_t P = c->P, Pm = c->Pm;
    CK(hipMalloc((void**)&c->zbuf_data, P * 8));
    CK(hipMemsetAsync(c->zbuf_data, 0xFF, P * 8, c->stream));
    CK(hipMalloc((void**)&c->eroded, P * sizeof(float4)));
    CK(hipMalloc((void**)&c->radius_conf, P * sizeof(float4)));
    CK(hipMemsetAsync(c->radius_conf, 0, P * sizeof(float4), c->stream));
    CK(hipMalloc((void**)&c->pixrec, P * 4 * sizeof(float4)));
    CK(hipMalloc((void**)&c->integrated, P));
    CK(hipMemsetAsync(c->integrated, 0, P, c->stream));
    CK(hipMalloc((void**)&c->extract_flags, (size_t)params->max_surfels));
    c->flagged.valid = false;
    CK(hipMalloc((void**)&c->index_map, P * 4));
    CK(hipMemsetAsync(c->index_map, 0, P * 4, c->stream));
    CK(hipMalloc((void**)&c->zbuf_a, Pm * 8));
    CK(hipMalloc((void**)&c->zbuf_b, Pm * 8));
    CK(hipMemsetAsync(c->zbuf_a, 0xFF, Pm * 8, c->stream));
    CK(hipMemsetAsync(c->zbuf_b, 0xFF, Pm * 8, c->stream));
    for (int b = 0; b < 2; ++b) CK(hipMalloc((void**)&c->surfels[b], (size_t)params->max_surfels * sizeof(suma_surfel)));
    CK(hipMalloc((void**)&c->poses, (size_t)params->max_poses * 16 * sizeof(float)));
    CK(hipMalloc((void**)&c->poses_inv, (size_t)params->max_poses * 16 * sizeof(float)));
    c->n_tiles_cap = (uint32_t)(((size_t)params->max_surfels + 2 * P) / SUMA_TILE + 2);
    CK(hipMalloc((void**)&c->tile_status, (size_t)c->n_tiles_cap * 8));
    CK(hipMemsetAsync(c->tile_status, 0, (size_t)c->n_tiles_cap * 8, c->stream));
    c->group_words = c->n_tiles_cap / 64 + 2;
    CK(hipMalloc((void**)&c->tile_group, (size_t)2 * c->group_words * 8));
    CK(hipMemsetAsync(c->tile_group, 0, (size_t)2 * c->group_words * 8, c->stream));
    CK(hipMalloc((void**)&c->ds, sizeof(DevState)));
    CK(hipHostMalloc((void**)&c->h_ds, sizeof(DevState), hipHostMallocDefault));
    memset(c->h_ds, 0, sizeof(DevState));
    /* ICP */
    c->icp_blocks = 256;
    CK(hipMalloc((void**)&c->gn, 2 * SUMA_MAX_HYP * sizeof(GnState)));
    CK(hipMemsetAsync(c->gn, 0, 2 * SUMA_MAX_HYP * sizeof(GnState), c->stream));
    CK(hipMalloc((void**)&c->gn_partial, (size_t)2 * SUMA_MAX_HYP * c->icp_blocks * SUMA_ACC_WORDS * sizeof(int64_t)));
    c->gn_part_launch = 0;
    c->gn_part_dirty[0] = c->gn_part_dirty[1] = c->gn_part_dirty[2] = 0;
    /* the rotating accumulator records of the Gauss-Newton chain start out zero (k_icp.hip, IterArgs) */
    CK(hipMemset(c->gn_partial, 0, (size_t)2 * SUMA_MAX_HYP * c->icp_blocks * SUMA_ACC_WORDS * sizeof(int64_t)));
    c->gn_launch = 0;
    c->gn_history_cap = 1025;
    CK(hipMalloc((void**)&c->gn_history, (size_t)c->gn_history_cap * 16 * sizeof(double)));
    CK(hipMalloc((void**)&c->gn_T0s, (size_t)SUMA_MAX_HYP * 16 * sizeof(double)));
    CK(hipHostMalloc((void**)&c->h_gn, SUMA_MAX_HYP * sizeof(GnState), hipHostMallocDefault));
    CK(hipMalloc((void**)&c->pose_block, 32 * sizeof(float)));
    CK(hipHostMalloc((void**)&c->h_rec, 2 * sizeof(HostResult), hipHostMallocDefault));
    memset(c->h_rec, 0, 2 * sizeof(HostResult));
    c->rec_seq = 0;
    c->gn_host_full = 0;
    c->gn_emit_pose = 0;
    c->gn_host_out = nullptr;
    c->gn_fused_report = nullptr;
    c->gn_fuse_k8 = 0;
    c->k8_fused_frame = nullptr;
    c->gn_host_seq = 0;
    c->cache_slots = nullptr;
    /* submap cache arena */
    /* default 16 x max_surfels (4.3 GB at the reference's 4.19 M): every tile of a KITTI-length
     * trajectory stays parked in HBM; re-extracted tiles take fresh arena space */
    uint64_t cache = params->cache_surfels ? params->cache_surfels : 16ull * params->max_surfels;
    if (cache > 0xffffffffull) cache = 0xffffffffull;
    c->cache_cap = (uint32_t)cache;
    CK(hipMalloc((void**)&c->cache_arena, (size_t)c->cache_cap * sizeof(suma_surfel)));
    c->cache_compactions = 0;
    c->cache_bound = 0;
    c->cache_slots_cap = 65536;
    CK(hipMalloc((void**)&c->cache_slots, (size_t)c->cache_slots_cap * sizeof(CacheSlot)));
    int r = frame_create_raw(c, params->model_width, params->model_height, &c->old_frame);
    if (r) return r;
    r = frame_create_raw(c, params->model_width, params->model_height, &c->new_frame);
    if (r) return r;
    r = frame_create_raw(c, params->model_width, params->model_height, &c->composed_frame);
    if (r) return r;
    r = map_reset_impl(c);
    if (r) return r;
    CK(hipStreamSynchronize(c->stream));
    return SUMA_OK;
  };
  rc = body();
  if (rc != SUMA_OK) {
    g_create_error = c->err;
    suma_ctx_destroy(c);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_live_ctx.insert(c);
  }
  *out = c;
  return SUMA_OK;
}

extern "C" void suma_ctx_destroy(suma_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_live_ctx.erase(c);
  }
  hipSetDevice(c->device);
  ingest_destroy(c);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->side_stream) {
    hipStreamSynchronize(c->side_stream);
    side_stream_released(c);
    hipStreamDestroy(c->side_stream);
    if (c->pre_event) hipEventDestroy(c->pre_event);
    if (c->order_event) hipEventDestroy(c->order_event);
  }
  if (c->h_rec) hipHostFree(c->h_rec);
  for (auto& ev : c->prof_events) {
    hipEventDestroy(ev.a);
    hipEventDestroy(ev.b);
  }
  for (auto& e : c->prof_pool) hipEventDestroy(e);
  void* dev[] = {c->extract_flags, c->pose_block, c->pixrec, c->zbuf_data, c->eroded,      c->radius_conf, c->integrated,  c->index_map, c->zbuf_a,
                 c->zbuf_b,    c->surfels[0],  c->surfels[1],  c->poses,       c->poses_inv, c->tile_status, c->tile_group,
                 c->ds,        c->gn,          c->gn_partial,  c->gn_history,  c->gn_T0s,    c->cache_arena,
                 c->cache_slots, c->scan_points, c->scan_labels, c->scan_probs, c->sync_flags, c->zbuf_k1,
                 c->filt_temp, c->filt_sort, c->filt_sort_tmp};
  for (void* p : dev)
    if (p) hipFree(p);
  if (c->h_ds) hipHostFree(c->h_ds);
  if (c->h_gn) hipHostFree(c->h_gn);
  suma_frame_destroy(c->old_frame);
  suma_frame_destroy(c->new_frame);
  suma_frame_destroy(c->composed_frame);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int suma_set_params(suma_ctx* c, const suma_params* p) {
  if (!c || !p) return SUMA_ERR_INVALID;
  if (p->data_width != c->p.data_width || p->data_height != c->p.data_height || p->model_width != c->p.model_width ||
      p->model_height != c->p.model_height || p->max_surfels != c->p.max_surfels || p->max_poses != c->p.max_poses)
    return fail(c, SUMA_ERR_INVALID, "suma_set_params: image sizes and capacities are fixed at creation");
  if (const char* fe = filter_params_error(p)) {
    c->err = std::string("suma_set_params: ") + fe;
    return SUMA_ERR_INVALID;
  }
  uint32_t cache = c->p.cache_surfels;
  c->p = *p;
  c->p.cache_surfels = cache;
  derive(c);
  c->params_version++;
  /* a Frame2Model object's own values (suma_icp_set_objective) do not outlive a new parameter block: whoever sends
   * parameters expects the next launch to use them; the adapter objects re-send theirs before every launch anyway */
  c->obj_set = false;
  return SUMA_OK;
}
extern "C" int suma_synchronize(suma_ctx* c) {
  if (!c) return SUMA_ERR_INVALID;
  if (c->side_stream) CK(hipStreamSynchronize(c->side_stream));
  CK(hipStreamSynchronize(c->stream));
  return SUMA_OK;
}
extern "C" void* suma_ctx_stream(suma_ctx* c) { return c ? (void*)c->stream : nullptr; }

/* ---------------------------------------------------------------------------------------------
 * frames
 * ------------------------------------------------------------------------------------------- */
extern "C" int suma_frame_create(suma_ctx* c, uint32_t w, uint32_t h, suma_frame** out) {
  if (!c || !out || w == 0 || h == 0) return SUMA_ERR_INVALID;
  return frame_create_raw(c, w, h, out);
}
extern "C" void suma_frame_destroy(suma_frame* f) {
  if (!f) return;
  /* nothing the context remembers by frame POINTER may outlive the frame (a new frame can get the same address and,
   * at timestamp 0, the same version: round-4 advisor) */
  if (suma_ctx* c = ctx_alive(f->ctx) ? f->ctx : nullptr) {
    if (c->k8_fused_frame == f) c->k8_fused_frame = nullptr;
    if (c->gate_frame == f) c->gate_frame = nullptr;
    if (c->rendered.out == f) c->rendered.valid = false;
  }
  if (f->map[0]) hipFree(f->map[0]);
  delete f;
}
extern "C" int suma_frame_copy(suma_ctx* c, suma_frame* dst, const suma_frame* src) {
  if (!c || !dst || !src || dst->width != src->width || dst->height != src->height) return SUMA_ERR_INVALID;
  size_t bytes = 3 * (size_t)src->width * src->height * sizeof(float4);
  if (c->gate_pending) CK(flush_gate(c));
  dst->version++;
  accessed(c, dst);
  accessed(c, src);
  CK(hipMemcpyAsync(dst->map[0], src->map[0], bytes, hipMemcpyDeviceToDevice, c->stream));
  return SUMA_OK;
}
extern "C" int suma_frame_download(suma_ctx* c, const suma_frame* f, int which, suma_float4* host) {
  if (!c || !f || !host || which < 0 || which > 2) return SUMA_ERR_INVALID;
  size_t bytes = (size_t)f->width * f->height * sizeof(float4);
  if (c->gate_pending) CK(flush_gate(c));
  CK(hipMemcpyAsync(host, f->map[which], bytes, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  host_synced(c);
  return SUMA_OK;
}
extern "C" int suma_frame_touch(suma_ctx* c, suma_frame* f) {
  if (!c || !f) return SUMA_ERR_INVALID;
  f->version++;
  return SUMA_OK;
}
extern "C" int suma_frame_upload(suma_ctx* c, suma_frame* f, int which, const suma_float4* host) {
  if (!c || !f || !host || which < 0 || which > 2) return SUMA_ERR_INVALID;
  size_t bytes = (size_t)f->width * f->height * sizeof(float4);
  if (c->gate_pending) CK(flush_gate(c));
  f->version++;
  CK(hipMemcpyAsync(f->map[which], host, bytes, hipMemcpyHostToDevice, c->stream));
  CK(hipStreamSynchronize(c->stream));
  host_synced(c);
  return SUMA_OK;
}
extern "C" uint32_t suma_frame_width(const suma_frame* f) { return f ? f->width : 0; }
extern "C" uint32_t suma_frame_height(const suma_frame* f) { return f ? f->height : 0; }
extern "C" void* suma_frame_device_ptr(const suma_frame* f, int which) {
  return (f && which >= 0 && which <= 2) ? (void*)f->map[which] : nullptr;
}

extern "C" int suma_frame_swap(suma_ctx* c, suma_frame* a, suma_frame* b) {
  if (!c || !a || !b) return SUMA_ERR_INVALID;
  if (a->width != b->width || a->height != b->height) return fail(c, SUMA_ERR_INVALID, "suma_frame_swap: sizes differ");
  /* the hazard bookkeeping follows the BUFFERS: a pending side-stream hand-off into either frame is flushed first, and
   * both handles inherit the later of the two ctx-stream accesses (round-4 advisor) */
  if (c->gate_pending && (c->gate_frame == a || c->gate_frame == b)) CK(flush_gate(c));
  for (int m = 0; m < 3; ++m) std::swap(a->map[m], b->map[m]);
  const uint64_t la = a->last_access > b->last_access ? a->last_access : b->last_access;
  a->last_access = b->last_access = la;
  a->version++;
  b->version++;
  return SUMA_OK;
}
extern "C" int suma_frame_export(suma_ctx* c, const suma_frame* f, int which, void** d_ptr, uint32_t* width,
                                 uint32_t* height, uint32_t* row_bytes) {
  if (!c || !f || !d_ptr || which < 0 || which > 2) return SUMA_ERR_INVALID;
  if (c->gate_pending) CK(flush_gate(c)); /* a consumer ordered behind the ctx stream sees the finished frame */
  *d_ptr = (void*)f->map[which];
  if (width) *width = f->width;
  if (height) *height = f->height;
  if (row_bytes) *row_bytes = f->width * (uint32_t)sizeof(float4);
  return SUMA_OK;
}

extern "C" int suma_device_alloc(suma_ctx* c, uint64_t bytes, void** d_ptr) {
  if (!c || !d_ptr) return SUMA_ERR_INVALID;
  CK(hipMalloc(d_ptr, bytes ? bytes : 16));
  return SUMA_OK;
}
extern "C" int suma_device_free(suma_ctx* c, void* d_ptr) {
  if (!c) return SUMA_ERR_INVALID;
  CK(hipFree(d_ptr));
  return SUMA_OK;
}
extern "C" int suma_device_upload(suma_ctx* c, void* d_dst, const void* host_src, uint64_t bytes) {
  if (!c || !d_dst || !host_src) return SUMA_ERR_INVALID;
  CK(hipMemcpyAsync(d_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return SUMA_OK;
}

extern "C" int suma_device_download(suma_ctx* c, void* host_dst, const void* d_src, uint64_t bytes) {
  if (!c || !host_dst || !d_src) return SUMA_ERR_INVALID;
  CK(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return SUMA_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Preprocessing::process
 * ------------------------------------------------------------------------------------------- */
extern "C" int suma_preprocess_device(suma_ctx* c, const suma_float4* d_points, const float* d_labels,
                                      const float* d_probs, uint32_t n, uint32_t timestamp, suma_frame* out) {
  if (!c || !out || (n > 0 && !d_points)) return SUMA_ERR_INVALID;
  if (out->width != c->p.data_width || out->height != c->p.data_height)
    return fail(c, SUMA_ERR_INVALID, "suma_preprocess: frame size differs from data_width x data_height");
  /* a second preprocessing while one is still pending on the side stream: keep them in order */
  if (c->gate_pending && c->ls == c->stream) CK(flush_gate(c));
  out->version++; /* the frame's maps are about to change (render de-duplication, fused K8 products) */
  if (c->ls == c->stream) accessed(c, out);
  CK(launch_preprocess(c, (const float4*)d_points, d_labels, d_probs, n, timestamp, out));
  return SUMA_OK;
}

static int stage_scan(suma_ctx* c, const suma_float4* points, const float* labels, const float* probs, uint32_t n) {
  if (n > c->scan_cap) {
    if (c->scan_points) hipFree(c->scan_points);
    if (c->scan_labels) hipFree(c->scan_labels);
    if (c->scan_probs) hipFree(c->scan_probs);
    c->scan_points = nullptr;
    c->scan_labels = c->scan_probs = nullptr;
    uint32_t cap = n + n / 4 + 1024;
    CK(hipMalloc((void**)&c->scan_points, (size_t)cap * sizeof(float4)));
    CK(hipMalloc((void**)&c->scan_labels, (size_t)cap * sizeof(float)));
    CK(hipMalloc((void**)&c->scan_probs, (size_t)cap * sizeof(float)));
    c->scan_cap = cap;
  }
  if (n == 0) return SUMA_OK;
  CK(hipMemcpyAsync(c->scan_points, points, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->ls));
  if (labels) CK(hipMemcpyAsync(c->scan_labels, labels, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->ls));
  if (probs) CK(hipMemcpyAsync(c->scan_probs, probs, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->ls));
  return SUMA_OK;
}

/* Preprocessing::process from the host vectors the reference's caller holds (Preprocessing.cpp:120-189 starts with
 * the glBufferData of Frame::points.assign): the scan is staged through pinned memory and the copy stream
 * (suma_ingest.hip) and K1-K3 run on the side stream behind the upload -- the call returns once the work is enqueued,
 * the render() that follows it in SurfelMapping::preprocess (SurfelMapping.cpp:344-351) does not read the frame and
 * overlaps it, and the first call that does read the frame waits on the device (flush_gate).  SUMA_PREPROCESS_SYNC=1
 * restores the pageable copy on the ctx stream. */
extern "C" int suma_preprocess(suma_ctx* c, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, uint32_t timestamp, suma_frame* out) {
  if (!c || !out || (n > 0 && !points)) return SUMA_ERR_INVALID;
  static const bool plain = getenv("SUMA_PREPROCESS_SYNC") != nullptr;
  if (plain) {
    int r = stage_scan(c, points, labels, probs, n);
    if (r) return r;
    return suma_preprocess_device(c, (const suma_float4*)c->scan_points, labels ? c->scan_labels : nullptr,
                                  probs ? c->scan_probs : nullptr, n, timestamp, out);
  }
  if (c->gate_pending) CK(flush_gate(c)); /* one hand-off at a time */
  int r = ensure_side_stream(c);
  if (r) return r;
  const suma_float4* dp;
  const float *dl, *dq;
  hipEvent_t up;
  void* slot;
  r = ingest_stage_blocking(c, points, labels, probs, n, &dp, &dl, &dq, &up, &slot);
  if (r) return r;
  if (c->side_stream) {
    /* the side stream must not overwrite the frame while earlier ctx-stream work still touches it: only if the
     * frame's last access on the ctx stream is younger than the last completion the host has observed (in
     * SurfelMapping::processScan it never is: the frame is the one of two scans ago, and the host has waited for a
     * minimisation since) is the side stream ordered behind the ctx stream -- otherwise K1-K3 overlap the
     * previous scan's surfel passes, as in the scan pipeline. */
    if (out->last_access > c->done_seq) {
      CK(hipEventRecord(c->order_event, c->stream));
      CK(hipStreamWaitEvent(c->side_stream, c->order_event, 0));
    }
    CK(hipStreamWaitEvent(c->side_stream, up, 0));
    c->ls = c->side_stream;
    r = suma_preprocess_device(c, dp, dl, dq, n, timestamp, out);
    c->ls = c->stream;
    ingest_consumed(c, slot, c->side_stream);
    if (r) return r;
    return side_handoff(c, out);
  }
  CK(hipStreamWaitEvent(c->stream, up, 0));
  r = suma_preprocess_device(c, dp, dl, dq, n, timestamp, out);
  ingest_consumed(c, slot, c->stream);
  return r;
}

/* ---------------------------------------------------------------------------------------------
 * Frame2Model + LieGaussNewton
 * ------------------------------------------------------------------------------------------- */
extern "C" int suma_icp_set_data(suma_ctx* c, const suma_frame* current, const suma_frame* model) {
  if (!c || !current || !model) return SUMA_ERR_INVALID;
  c->icp_current = current;
  c->icp_model = model;
  c->icp_iteration0 = 0; /* iteration_ = 0, Frame2Model.cpp:122 */
  return SUMA_OK;
}

extern "C" int suma_icp_set_objective(suma_ctx* c, const suma_icp_objective* o) {
  if (!c) return SUMA_ERR_INVALID;
  c->obj_set = (o != nullptr);
  if (o) c->obj = *o;
  return SUMA_OK;
}
/* LieGaussNewton::information(): JtJ of the last step, as left in the pinned state copy by the last
 * suma_icp_minimize / suma_icp_jacobian_products */
extern "C" int suma_icp_information(suma_ctx* c, double information[36]) {
  if (!c || !information) return SUMA_ERR_INVALID;
  memcpy(information, c->h_gn[0].JtJ, 36 * sizeof(double));
  return SUMA_OK;
}

static void fill_stats(const GnState& g, suma_icp_stats* st) {
  if (!st) return;
  st->error = g.F;
  st->inlier_residual = g.F_inlier;
  st->valid = g.valid;
  st->outlier = g.outlier;
  st->inlier = g.valid - g.outlier;
  st->invalid = g.invalid;
  st->iterations = g.k;
  st->converged = g.converged;
}

static void fill_stats_host(const HostResult& g, suma_icp_stats* st) {
  if (!st) return;
  st->error = g.F;
  st->inlier_residual = g.F_inlier;
  st->valid = g.valid;
  st->outlier = g.outlier;
  st->inlier = g.valid - g.outlier;
  st->invalid = g.invalid;
  st->iterations = g.k;
  st->converged = g.converged;
}

/* Poll a HostResult until the closing launch has stamped it.  The stream is queried now and then so that
 * a faulted or drained stream surfaces as an error instead of an endless spin. */
static int wait_host_result(suma_ctx* c, const HostResult* h, uint32_t seq) {
  for (uint32_t spins = 1;; ++spins) {
    if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == seq) return SUMA_OK;
    __builtin_ia32_pause();
    if ((spins & 0xfffu) == 0) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) { /* everything has run: the record must be there */
        if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == seq) return SUMA_OK;
        return fail(c, SUMA_ERR_HIP, "minimisation result was not reported");
      }
      if (q != hipErrorNotReady) CK(q);
    }
  }
}

/* host side of a HostResult.acc report: JtJ (6x6 column-major, mirrored from the packed upper triangle), Jtr and the
 * GnState copy suma_icp_information reads -- the same value the device forms, (double)word * 2^-28 */
static void unpack_acc(suma_ctx* c, const HostResult& h, double* JtJ, double* Jtr, int64_t* acc) {
  GnState& g = c->h_gn[0];
  for (int w = 0; w < (int)SUMA_ACC_WORDS; ++w) g.acc[w] = h.acc[w];
  for (int k = 0; k < 36; ++k) {
    const int i = k % 6, j = k / 6, lo = i < j ? i : j, hi = i < j ? j : i;
    g.JtJ[k] = (double)h.acc[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)] * (1.0 / SUMA_ACC_SCALE);
  }
  for (int k = 0; k < 6; ++k) g.Jtr[k] = (double)h.acc[21 + k] * (1.0 / SUMA_ACC_SCALE);
  if (JtJ) memcpy(JtJ, g.JtJ, sizeof(g.JtJ));
  if (Jtr) memcpy(Jtr, g.Jtr, sizeof(g.Jtr));
  if (acc) memcpy(acc, g.acc, sizeof(g.acc));
}

extern "C" int suma_icp_jacobian_products(suma_ctx* c, const double pose[16], uint32_t iteration, double JtJ[36],
                                          double Jtr[6], int64_t* acc, suma_icp_stats* stats) {
  if (!c || !pose) return SUMA_ERR_INVALID;
  if (!c->icp_current || !c->icp_model) return fail(c, SUMA_ERR_INVALID, "suma_icp_set_data has not been called");
  if (c->gate_pending) CK(flush_gate(c));
  accessed(c, c->icp_current);
  accessed(c, c->icp_model);
  CK(launch_gn_init(c, pose, 1, 0, iteration));
  /* ONE launch: the pixel pass closes itself (its last block totals the accumulator records) and reports the sums
   * straight into a pinned host record the host polls -- no consume-only launch, no copy command, no stream
   * synchronisation (round 3: two launches + hipMemcpyAsync + hipStreamSynchronize).  If the frame is data-sized and
   * its K8 products (init_radiusConf.vert: pose independent) do not exist yet for this timestamp, the pass forms them
   * on the texels it streams anyway, as the scan pipeline's statistics pass does: the reference evaluates the
   * objective on currentFrame_ right before updateMap (SurfelMapping.cpp:411-413 -> :799). */
  const suma_frame* cur = c->icp_current;
  const bool k8_wanted = cur->width == c->p.data_width && cur->height == c->p.data_height &&
                         !(c->k8_fused_frame == cur && c->k8_fused_version == cur->version &&
                           c->k8_fused_stamp == c->timestamp && c->k8_fused_params == c->params_version);
  HostResult* rec = &c->h_rec[1];
  c->rec_seq += 1;
  {
    ProfScope ps(c, k8_wanted ? "k6k8_stats_radius" : "k6_icp_step", (96.0 + (k8_wanted ? 81.0 : 0.0)) * (double)cur->width * cur->height);
    c->gn_fused_report = rec;
    c->gn_host_seq = c->rec_seq;
    c->gn_host_full = 1;
    c->gn_fuse_k8 = k8_wanted ? 1 : 0;
    hipError_t e = launch_icp_iteration(c, 1, 1, 0.0, 0.0, 1, 0, 1);
    c->gn_fuse_k8 = 0;
    c->gn_host_full = 0;
    c->gn_fused_report = nullptr;
    CK(e);
  }
  if (k8_wanted) {
    c->k8_fused_frame = cur;
    c->k8_fused_version = cur->version;
    c->k8_fused_stamp = c->timestamp;
    c->k8_fused_params = c->params_version;
  }
  const uint64_t mark = c->enq_seq;
  int r = wait_host_result(c, rec, c->rec_seq);
  if (r) return r;
  c->done_seq = mark;
  unpack_acc(c, *rec, JtJ, Jtr, acc);
  fill_stats_host(*rec, stats);
  if (stats) stats->iterations = 0;
  return SUMA_OK;
}

/* enqueue one whole minimisation.  max iterations > 0: no host round trip inside.  max iterations == 0 means
 * "iterate until convergence" in the reference (LieGaussNewton.cpp:27): launches are enqueued in chunks and the
 * chains' done flags polled in between, so 0 really runs until every chain has converged; a chain that has not
 * converged after SUMA_GN_HARD_CAP iterations is reported as an error instead of returned silently. */
#define SUMA_GN_CHUNK 32u
#define SUMA_GN_HARD_CAP (1u << 16)
static int enqueue_minimize(suma_ctx* c, const double* T0s, uint32_t n_hyp, int with_history, uint32_t iteration0 = 0,
                            uint32_t iteration0_rest = 0) {
  const uint32_t max_iter = c->p.max_iterations;
  const uint32_t iter_arg = max_iter > 0 ? max_iter : 0xffffffffu;
  if (c->gate_pending) CK(flush_gate(c)); /* the chain reads the frame the side stream preprocessed (k_sync.hip) */
  accessed(c, c->icp_current);
  accessed(c, c->icp_model);
  if (with_history) c->hist_seq += 1; /* this chain overwrites the device-side pose history */
  CK(launch_gn_init(c, T0s, n_hyp, with_history, iteration0, iteration0_rest));
  /* launch j runs the pixel phase of iteration j after consuming the sums of iteration j-1; the
   * closing launch only consumes */
  const double launch_bytes = 96.0 * (double)c->icp_current->width * c->icp_current->height * n_hyp;
  if (max_iter > 0) {
    /* one event pair around the whole chain of identical pixel launches: per-launch time = chain / N */
    ProfScope ps(c, "k6_icp_step", launch_bytes * max_iter, max_iter);
    for (uint32_t i = 0; i < max_iter; ++i)
      CK(launch_icp_iteration(c, n_hyp, iter_arg, (double)c->p.stopping_threshold, (double)c->p.delta, 0, with_history, 1));
  } else {
    uint32_t total = 0;
    for (;;) {
      {
        ProfScope ps(c, "k6_icp_step", launch_bytes * SUMA_GN_CHUNK, SUMA_GN_CHUNK);
        for (uint32_t i = 0; i < SUMA_GN_CHUNK; ++i)
          CK(launch_icp_iteration(c, n_hyp, iter_arg, (double)c->p.stopping_threshold, (double)c->p.delta, 0, with_history, 1));
      }
      total += SUMA_GN_CHUNK;
      CK(hipMemcpyAsync(c->h_gn, gn_result(c), (size_t)n_hyp * sizeof(GnState), hipMemcpyDeviceToHost, c->stream));
      CK(hipStreamSynchronize(c->stream));
      host_synced(c);
      bool all_done = true;
      for (uint32_t h = 0; h < n_hyp; ++h) all_done = all_done && c->h_gn[h].done;
      if (all_done) break;
      if (total >= SUMA_GN_HARD_CAP)
        return fail(c, SUMA_ERR_INVALID, "Gauss-Newton did not converge within 65536 iterations (max iterations = 0)");
    }
  }
  {
    ProfScope ps(c, "k6_icp_finish", 0.0);
    CK(launch_icp_iteration(c, n_hyp, iter_arg, (double)c->p.stopping_threshold, (double)c->p.delta, 0, with_history, 0));
  }
  return SUMA_OK;
}

/* The closing launch of the chain reports into a pinned host record (pose, statistics, information matrix, history
 * length) that the host polls: no copy command and no stream synchronisation behind the minimisation (round 3:
 * hipMemcpyAsync + hipStreamSynchronize, and a second pair for the history).  The pose history is always recorded on
 * the device; it crosses to the host only when asked for -- here, or later through suma_icp_history. */
extern "C" int suma_icp_minimize(suma_ctx* c, const double T0[16], double T_out[16], double* history,
                                 uint32_t history_cap, uint32_t* n_hist, suma_icp_stats* stats) {
  if (!c || !T0 || !T_out) return SUMA_ERR_INVALID;
  if (!c->icp_current || !c->icp_model) return fail(c, SUMA_ERR_INVALID, "suma_icp_set_data has not been called");
  HostResult* rec = &c->h_rec[0];
  c->rec_seq += 1;
  c->gn_host_out = rec;
  c->gn_host_seq = c->rec_seq;
  c->gn_host_full = 1;
  const uint32_t iteration0 = c->icp_iteration0; /* suma_icp_set_iteration, one shot */
  c->icp_iteration0 = 0;
  int r = enqueue_minimize(c, T0, 1, 1, iteration0);
  c->gn_host_out = nullptr;
  c->gn_host_full = 0;
  if (r) return r;
  const uint64_t mark = c->enq_seq;
  r = wait_host_result(c, rec, c->rec_seq);
  if (r) return r;
  c->done_seq = mark;
  memcpy(T_out, rec->Tk, sizeof(rec->Tk));
  fill_stats_host(*rec, stats);
  memcpy(c->h_gn[0].JtJ, rec->JtJ, sizeof(rec->JtJ)); /* suma_icp_information */
  memcpy(c->h_gn[0].Jtr, rec->Jtr, sizeof(rec->Jtr));
  c->last_n_hist = rec->n_hist;
  if (n_hist) *n_hist = rec->n_hist;
  if (history != nullptr && history_cap > 0) return suma_icp_history(c, history, history_cap, nullptr);
  return SUMA_OK;
}

extern "C" int suma_icp_set_iteration(suma_ctx* c, uint32_t iteration) {
  if (!c) return SUMA_ERR_INVALID;
  c->icp_iteration0 = iteration;
  return SUMA_OK;
}

extern "C" uint64_t suma_icp_history_sequence(const suma_ctx* c) { return c ? c->hist_seq : 0; }

extern "C" int suma_icp_history(suma_ctx* c, double* history, uint32_t history_cap, uint32_t* n_hist) {
  if (!c || (!history && history_cap)) return SUMA_ERR_INVALID;
  if (n_hist) *n_hist = c->last_n_hist;
  uint32_t n = c->last_n_hist < history_cap ? c->last_n_hist : history_cap;
  if (n > c->gn_history_cap) n = c->gn_history_cap;
  if (n) {
    CK(hipMemcpyAsync(history, c->gn_history, (size_t)n * 16 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream));
    host_synced(c);
  }
  return SUMA_OK;
}

extern "C" int suma_icp_minimize_batch(suma_ctx* c, const double* T0s, uint32_t n_hyp, double* T_out,
                                       suma_icp_stats* stats) {
  if (!c || !T0s || !T_out || n_hyp == 0 || n_hyp > SUMA_MAX_HYP) return SUMA_ERR_INVALID;
  if (!c->icp_current || !c->icp_model) return fail(c, SUMA_ERR_INVALID, "suma_icp_set_data has not been called");
  int r = enqueue_minimize(c, T0s, n_hyp, 0);
  if (r) return r;
  CK(hipMemcpyAsync(c->h_gn, gn_result(c), (size_t)n_hyp * sizeof(GnState), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  host_synced(c);
  for (uint32_t h = 0; h < n_hyp; ++h) {
    memcpy(T_out + 16 * (size_t)h, c->h_gn[h].Tk, 16 * sizeof(double));
    if (stats) fill_stats(c->h_gn[h], &stats[h]);
  }
  return SUMA_OK;
}

/* ---------------------------------------------------------------------------------------------
 * SurfelMap
 * ------------------------------------------------------------------------------------------- */
extern "C" int suma_map_reset(suma_ctx* c) {
  if (!c) return SUMA_ERR_INVALID;
  return map_reset_impl(c);
}

static void submap_center(const suma_ctx* c, int32_t i, int32_t j, float* cx, float* cy) {
  *cx = (float)(2.0 * i * c->p.submap_extent); /* SurfelMap.cpp:704-706 */
  *cy = (float)(2.0 * j * c->p.submap_extent);
}

static int cache_slot_for(suma_ctx* c, int32_t i, int32_t j, uint32_t* slot) {
  auto key = std::make_pair(i, j);
  auto it = c->cache_index.find(key);
  if (it != c->cache_index.end()) {
    *slot = it->second;
    return SUMA_OK;
  }
  uint32_t s = (uint32_t)c->cache_index.size();
  if (s >= c->cache_slots_cap) return fail(c, SUMA_ERR_CAPACITY, "submap cache slot table exhausted");
  c->cache_index[key] = s;
  *slot = s;
  return SUMA_OK;
}

/* The cache arena is a bump allocator (K12's finaliser moves DevState.cache_used): a tile that is extracted AGAIN gets a
 * new block and leaves its old one behind, where the reference simply overwrites the tile's std::vector
 * (SurfelMap.cpp:733-734).  When the arena is about to run full, the live blocks are copied, in slot order, into a fresh
 * arena and the old one is freed -- rare (every few thousand scans at the default size), so a synchronous host-driven
 * pass is fine; the stale blocks are what is reclaimed.  Returns SUMA_OK when the arena has room for one more tile. */
static int cache_compact_if_needed(suma_ctx* c, uint32_t pending_slot) {
  /* the host's view of the bump pointer lags the device by at most one update (one tile, <= SUMA_EXTRACT_CAPACITY) */
  /* room wanted for the next tile: the reference's per-tile capacity (SurfelMap.cpp:279), or an eighth of a small arena;
   * whether the tile really fits is K12's own check (DevState.overflow bit 1) */
  const uint64_t need = SUMA_EXTRACT_CAPACITY < c->cache_cap / 8 ? SUMA_EXTRACT_CAPACITY : c->cache_cap / 8;
  /* the bump pointer lives on the device; the host keeps an upper bound (exact value at the last read-back + what
   * every extraction since can have added at most) and synchronises only when the bound gets close */
  if (c->cache_bound + need <= c->cache_cap) return SUMA_OK;
  /* the last attempt found every allocated block live: K12's own capacity check (DevState.overflow bit 1) decides;
   * synchronising and reading the slot table again before every extraction would buy nothing */
  if (c->cache_nothing_stale) return SUMA_OK;
  int r = read_state(c); /* synchronises; the exact pointer */
  if (r) return r;
  host_synced(c);
  c->cache_bound = c->h_ds->cache_used;
  if (c->cache_bound + need <= c->cache_cap) return SUMA_OK;
  const uint32_t ns = (uint32_t)c->cache_index.size();
  std::vector<CacheSlot> slots(ns);
  if (ns) CK(hipMemcpy(slots.data(), c->cache_slots, ns * sizeof(CacheSlot), hipMemcpyDeviceToHost));
  /* the slot of the extraction that is about to be enqueued has been assigned but not written by K12 yet: whatever
   * the table holds for it (a tile that is being re-extracted: its old block) is stale by definition */
  if (pending_slot < ns) slots[pending_slot].count = 0;
  uint64_t live = 0;
  for (auto& q : slots) live += q.count;
  if (live >= c->h_ds->cache_used) { /* nothing stale to reclaim */
    c->cache_nothing_stale = true;
    return SUMA_OK;
  }
  suma_surfel* fresh = nullptr;
  CK(hipMalloc((void**)&fresh, (size_t)c->cache_cap * sizeof(suma_surfel)));
  uint32_t off = 0;
  hipError_t e = hipSuccess;
  for (auto& q : slots) {
    if (q.count && e == hipSuccess)
      e = hipMemcpyAsync(fresh + off, c->cache_arena + q.offset, (size_t)q.count * sizeof(suma_surfel),
                         hipMemcpyDeviceToDevice, c->stream);
    q.offset = off;
    off += q.count;
  }
  if (ns && e == hipSuccess) e = hipMemcpyAsync(c->cache_slots, slots.data(), ns * sizeof(CacheSlot), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&c->ds->cache_used, &off, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    hipFree(fresh); /* the old arena and its table stay in place */
    CK(e);
  }
  hipFree(c->cache_arena);
  c->cache_arena = fresh;
  c->h_ds->cache_used = off;
  c->cache_bound = off;
  c->cache_compactions += 1;
  c->cache_nothing_stale = false;
  return SUMA_OK;
}

/* SurfelMap::extractSurfels, SurfelMap.cpp:708-742 (tiles are popped from the back) */
static int extract_surfels(suma_ctx* c, bool partially) {
  while (!c->extraction.empty()) {
    auto idx = c->extraction.back();
    c->extraction.pop_back();
    float cx, cy;
    submap_center(c, idx.first, idx.second, &cx, &cy);
    const auto account = [&]() {
      const uint64_t most = (uint64_t)c->known_surfels + 2 * c->P; /* a tile holds at most the whole map ... */
      c->cache_bound += most < SUMA_EXTRACT_CAPACITY ? most : SUMA_EXTRACT_CAPACITY; /* ... or K12's capacity */
    };
    if (c->flagged.valid && c->flagged.fused && c->flagged.i == idx.first && c->flagged.j == idx.second) {
      /* the update that has just run wrote this tile's records to the arena and committed its slot */
      c->flagged.valid = false;
      account();
      if (partially) break;
      continue;
    }
    if (c->flagged.valid && c->flagged.fused)
      return fail(c, SUMA_ERR_INVALID, "internal: the update extracted a tile other than the one updateActiveSubmaps asks for");
    uint32_t slot;
    int r = cache_slot_for(c, idx.first, idx.second, &slot);
    if (r) return r;
    r = cache_compact_if_needed(c, slot);
    if (r) return r;
    /* K9 / K10 of the update that has just run flagged the surfels of this very tile at their final index
     * (peek_extraction): the extraction reads one byte per surfel instead of position + creation stamp of the whole map */
    const int use_flags = (c->flagged.valid && c->flagged.i == idx.first && c->flagged.j == idx.second) ? 1 : 0;
    c->flagged.valid = false;
    CK(launch_extract(c, slot, cx, cy, c->p.submap_extent, use_flags));
    account();
    if (partially) break;
  }
  return SUMA_OK;
}

static int append_cached(suma_ctx* c, int32_t i, int32_t j) {
  auto it = c->cache_index.find(std::make_pair(i, j));
  if (it == c->cache_index.end()) return SUMA_OK; /* never extracted: an empty SubmapCache */
  CK(launch_append_cached(c, it->second));
  return SUMA_OK;
}

/* The tile updateActiveSubmaps -> extractSurfels(partially = true) will extract right after the update at `pose`
 * (SurfelMap.cpp:744-824, 708-742): the same decisions on a copy of the state.  false: none, or not exactly one. */
static bool peek_extraction(const suma_ctx* c, const float* pose, int32_t* ti, int32_t* tj, bool* appends) {
  *appends = false;
  const auto cached = [&](int32_t i, int32_t j) { return c->cache_index.find(std::make_pair(i, j)) != c->cache_index.end(); };
  if (!c->p.partial_extraction || getenv("SUMA_NO_EXTRACT_FLAGS")) return false; /* all pending tiles in one go: no single tile to flag */
  const int32_t dim = c->p.submap_dimension;
  const float ext = c->p.submap_extent;
  int32_t oi = c->origin_i, oj = c->origin_j;
  float cx, cy;
  submap_center(c, oi, oj, &cx, &cy);
  const float changex = pose[12] - cx, changey = pose[13] - cy, factor = 1.1f;
  bool have = !c->extraction.empty();
  std::pair<int32_t, int32_t> last = have ? c->extraction.back() : std::make_pair(0, 0);
  if (fabsf(changex) > factor * ext) {
    const int32_t dir = (changex < 0) ? -1 : 1;
    last = {oi - dir * dim, oj + dim}; /* the last tile of the pushed row, k = dim */
    have = true;
    oi += dir;
    for (int32_t k = -dim; k <= dim; ++k) *appends |= cached(oi + dir * dim, oj + k);
  }
  if (fabsf(changey) > factor * ext) {
    const int32_t dir = (changey < 0) ? -1 : 1;
    last = {oi + dim, oj - dir * dim};
    have = true;
    for (int32_t r0 = -dim; r0 <= dim; ++r0) *appends |= cached(oi + r0, oj + dir + dir * dim);
  }
  if (!have) return false;
  *ti = last.first;
  *tj = last.second;
  return true;
}

/* SurfelMap::updateActiveSubmaps, SurfelMap.cpp:744-824 */
static int update_active_submaps(suma_ctx* c, const float* pose) {
  const int32_t dim = c->p.submap_dimension;
  const float ext = c->p.submap_extent;
  float cx, cy;
  submap_center(c, c->origin_i, c->origin_j, &cx, &cy);
  float changex = pose[12] - cx, changey = pose[13] - cy;
  const float factor = 1.1f;
  if (fabsf(changex) > factor * ext || fabsf(changey) > factor * ext) {
    if (fabsf(changex) > factor * ext) {
      int32_t dir = (changex < 0) ? -1 : 1;
      for (int32_t k = -dim; k <= dim; ++k) c->extraction.push_back({c->origin_i - dir * dim, c->origin_j + k});
      c->origin_i += dir;
      for (int32_t k = -dim; k <= dim; ++k) {
        int r = append_cached(c, c->origin_i + dir * dim, c->origin_j + k);
        if (r) return r;
      }
    }
    if (fabsf(changey) > factor * ext) {
      int32_t dir = (changey < 0) ? -1 : 1;
      for (int32_t r0 = -dim; r0 <= dim; ++r0) c->extraction.push_back({c->origin_i + r0, c->origin_j - dir * dim});
      c->origin_j += dir;
      for (int32_t r0 = -dim; r0 <= dim; ++r0) {
        int r = append_cached(c, c->origin_i + r0, c->origin_j + dir * dim);
        if (r) return r;
      }
    }
  }
  if (!c->extraction.empty()) return extract_surfels(c, c->p.partial_extraction != 0);
  return SUMA_OK;
}

extern "C" int suma_map_update(suma_ctx* c, const float pose[16], const suma_frame* frame) {
  if (c && c->gate_pending) CK(flush_gate(c));
  if (!c || !pose || !frame) return SUMA_ERR_INVALID;
  if (frame->width != c->p.data_width || frame->height != c->p.data_height)
    return fail(c, SUMA_ERR_INVALID, "suma_map_update: frame size differs from data_width x data_height");
  if (c->timestamp >= c->p.max_poses)
    return fail(c, SUMA_ERR_CAPACITY, "pose table full (max_poses; reference: maxPoses_ = 10000, SurfelMap.h:205)");
  /* SurfelMap.cpp:494-495 (poses_[timestamp_] = pose) is done by the first kernel of the update */
  float inv_pose[16];
  rigid_inverse_f(pose, inv_pose);
  /* K11 area, SurfelMap.cpp:667-677 */
  float cx, cy;
  submap_center(c, c->origin_i, c->origin_j, &cx, &cy);
  float extent = 2.0f * (float)c->p.submap_dimension * c->p.submap_extent + c->p.submap_extent;
  if (c->p.partial_extraction && !c->extraction.empty()) extent += 2.0f * c->p.submap_extent;
  int k7_done = 0;
  if (c->k7.valid) { /* zbuf_data holds an index-map splat made by the post-ICP render pass */
    k7_done = (c->k7.map_version == c->map_version && c->k7.params_version == c->params_version &&
               memcmp(c->k7.pose, pose, 16 * sizeof(float)) == 0 && frame->width == c->p.data_width);
    if (!k7_done) {
      CK(launch_clear_index_zbuf(c)); /* different pose after all (fallback ICP): redo K7 */
      c->k7_spec.on = false;          /* a speculated splat that was paid for and dropped */
    }
    c->k7.valid = false;
  }
  /* speculation of suma_map_render_active: on again as soon as an update arrives that matches the last active render */
  if (k7_done || (c->k7_spec.have_last && c->k7_spec.map_version == c->map_version &&
                  c->k7_spec.params_version == c->params_version &&
                  memcmp(c->k7_spec.last_pose, pose, 16 * sizeof(float)) == 0))
    c->k7_spec.on = true;
  c->k7_spec.have_last = false;
  accessed(c, frame);
  /* the K8 products of the fused statistics pass belong to the frame CONTENTS they were made from */
  if (c->k8_fused_frame == frame && c->k8_fused_version != frame->version) c->k8_fused_frame = nullptr;
  float ex[3];
  int32_t ti = 0, tj = 0;
  bool appends = false;
  const bool flag_tile = peek_extraction(c, pose, &ti, &tj, &appends);
  int fused_slot = -1;
  if (flag_tile) {
    submap_center(c, ti, tj, &ex[0], &ex[1]);
    ex[2] = c->p.submap_extent;
    /* no cached tile comes back into the map between this update and the extraction (the usual case: the sensor is not
     * revisiting): the extraction is the update's own stream-out (k9_update<true>, k10_generate<true>).  Anything that
     * stands in the way -- slot table full, arena to be compacted and still full -- is left to the extraction proper,
     * which reports it where it always did */
    if (!appends && !getenv("SUMA_NO_FUSED_EXTRACT") && c->cache_index.size() < c->cache_slots_cap) {
      uint32_t slot;
      if (cache_slot_for(c, ti, tj, &slot) == SUMA_OK && cache_compact_if_needed(c, slot) == SUMA_OK) fused_slot = (int)slot;
    }
  }
  c->flagged.valid = flag_tile;
  c->flagged.i = ti;
  c->flagged.j = tj;
  c->flagged.fused = fused_slot >= 0;
  c->flagged.slot = fused_slot >= 0 ? (uint32_t)fused_slot : 0u;
  CK(launch_map_update(c, pose, inv_pose, frame, cx, cy, extent, k7_done, flag_tile ? ex : nullptr, fused_slot));
  c->cur ^= 1;
  c->map_version++;
  int r = update_active_submaps(c, pose);
  if (r) return r;
  c->timestamp += 1;
  return SUMA_OK;
}

/* render() with de-duplication: if OLD / NEW / `out` already hold exactly this rendering (same
 * poses, threshold, parameters and map contents, nothing has overwritten them since), the launch
 * is skipped -- the result would be identical bit for bit. */
static int map_render_dedup(suma_ctx* c, const float* pose_old, const float* pose_new, float conf_threshold,
                            suma_frame* out) {
  auto& r = c->rendered;
  /* the three targets still hold that rendering: nothing has written them since (suma_frame.version is bumped by every
   * writing call, also for caller-owned frames -- round 3 gave up on those) */
  if (r.valid && r.out == out && r.map_version == c->map_version && r.params_version == c->params_version &&
      r.out_version == out->version && r.old_version == c->old_frame->version && r.new_version == c->new_frame->version &&
      memcmp(&r.conf_threshold, &conf_threshold, sizeof(float)) == 0 &&
      memcmp(r.pose_old, pose_old, 16 * sizeof(float)) == 0 && memcmp(r.pose_new, pose_new, 16 * sizeof(float)) == 0)
    return SUMA_OK;
  if (c->gate_pending && c->gate_frame == out) CK(flush_gate(c));
  CK(launch_map_render(c, pose_old, pose_new, conf_threshold, out));
  out->version++;
  if (c->old_frame != out) c->old_frame->version++;
  if (c->new_frame != out) c->new_frame->version++;
  accessed(c, out);
  accessed(c, c->old_frame);
  accessed(c, c->new_frame);
  r.out_version = out->version;
  r.old_version = c->old_frame->version;
  r.new_version = c->new_frame->version;
  r.valid = true;
  r.out = out;
  r.map_version = c->map_version;
  r.params_version = c->params_version;
  r.conf_threshold = conf_threshold;
  memcpy(r.pose_old, pose_old, 16 * sizeof(float));
  memcpy(r.pose_new, pose_new, 16 * sizeof(float));
  return SUMA_OK;
}

extern "C" int suma_map_render(suma_ctx* c, const float pose_old[16], const float pose_new[16], float conf_threshold,
                               suma_frame* out) {
  if (!c || !pose_old || !pose_new || !out) return SUMA_ERR_INVALID;
  if (out->width != c->p.model_width || out->height != c->p.model_height)
    return fail(c, SUMA_ERR_INVALID, "suma_map_render: frame size differs from model_width x model_height");
  return map_render_dedup(c, pose_old, pose_new, conf_threshold, out);
}
extern "C" int suma_map_render_active(suma_ctx* c, const float pose[16], float conf_threshold) {
  if (!c || !pose) return SUMA_ERR_INVALID;
  if (c->gate_pending && c->gate_frame == c->new_frame) CK(flush_gate(c));
  /* The reference's updatePose renders the active map at pose_new * increment (SurfelMapping.cpp:406) and updateMap
   * then calls update() with that very pose (:799), whose first pass is the index map of the same surfels from the
   * same pose (K7): the splat rides on this pass, as in the scan pipeline, and suma_map_update takes it if map,
   * parameters and pose still match (bit for bit) -- otherwise it is cleared and K7 runs as a pass of its own.  A splat
   * that was not consumed switches the speculation off until an update arrives that would have matched. */
  int fuse_k7 = 0;
  if (c->k7.valid) { /* the previous splat was never consumed */
    CK(launch_clear_index_zbuf(c));
    c->k7.valid = false;
    c->k7_spec.on = false;
  } else if (c->k7_spec.on && c->p.data_width == c->p.model_width && !getenv("SUMA_NO_K7_SPECULATION")) {
    fuse_k7 = 1;
  }
  CK(launch_map_render_single(c, pose, conf_threshold, 1, fuse_k7, nullptr));
  c->new_frame->version++;
  accessed(c, c->new_frame);
  if (fuse_k7) {
    c->k7.valid = true;
    c->k7.map_version = c->map_version;
    c->k7.params_version = c->params_version;
    memcpy(c->k7.pose, pose, 16 * sizeof(float));
  }
  c->k7_spec.have_last = true;
  c->k7_spec.map_version = c->map_version;
  c->k7_spec.params_version = c->params_version;
  memcpy(c->k7_spec.last_pose, pose, 16 * sizeof(float));
  return SUMA_OK;
}
extern "C" int suma_map_render_inactive(suma_ctx* c, const float pose[16], float conf_threshold) {
  if (!c || !pose) return SUMA_ERR_INVALID;
  if (c->gate_pending && c->gate_frame == c->old_frame) CK(flush_gate(c));
  CK(launch_map_render_single(c, pose, conf_threshold, 0, 0, nullptr));
  c->old_frame->version++;
  accessed(c, c->old_frame);
  return SUMA_OK;
}
extern "C" int suma_map_render_composed(suma_ctx* c, const float pose_old[16], const float pose_new[16],
                                        float conf_threshold) {
  if (!c || !pose_old || !pose_new) return SUMA_ERR_INVALID;
  if (c->gate_pending && c->gate_frame == c->composed_frame) CK(flush_gate(c));
  CK(launch_map_render_composed(c, pose_old, pose_new, conf_threshold)); /* touches COMPOSED only */
  c->composed_frame->version++;
  accessed(c, c->composed_frame);
  return SUMA_OK;
}
extern "C" suma_frame* suma_map_frame(suma_ctx* c, int which) {
  if (!c) return nullptr;
  return which == SUMA_FRAME_OLD ? c->old_frame : (which == SUMA_FRAME_NEW ? c->new_frame : c->composed_frame);
}

extern "C" int suma_map_update_poses(suma_ctx* c, const float* poses16, uint32_t n) {
  if (!c || (!poses16 && n)) return SUMA_ERR_INVALID;
  if (n > c->p.max_poses) n = c->p.max_poses;
  if (n == 0) return SUMA_OK;
  float* d_tmp = nullptr;
  CK(hipMalloc((void**)&d_tmp, (size_t)n * 16 * sizeof(float)));
  hipError_t e = hipMemcpyAsync(d_tmp, poses16, (size_t)n * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_set_poses(c, d_tmp, 0, n);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_tmp);
  c->map_version++;
  CK(e);
  return SUMA_OK;
}

static int check_overflow(suma_ctx* c) {
  if (c->h_ds->overflow & 1u) return fail(c, SUMA_ERR_CAPACITY, "surfel capacity (max_surfels) exceeded; map truncated");
  if (c->h_ds->overflow & 2u) return fail(c, SUMA_ERR_CAPACITY, "submap cache arena (cache_surfels) exhausted");
  if (c->h_ds->overflow & 8u) {
    char msg[160];
    snprintf(msg, sizeof(msg), "stable-compaction / stream hand-off timed out (internal error, site mask 0x%x)", c->h_ds->fault_site);
    return fail(c, SUMA_ERR_HIP, msg);
  }
  return SUMA_OK;
}

extern "C" int suma_map_size(suma_ctx* c, uint32_t* n) {
  if (!c || !n) return SUMA_ERR_INVALID;
  int r = read_state(c);
  if (r) return r;
  *n = c->h_ds->n_surfels;
  return check_overflow(c);
}
extern "C" int suma_map_timestamp(suma_ctx* c, uint32_t* t) {
  if (!c || !t) return SUMA_ERR_INVALID;
  *t = c->timestamp;
  return SUMA_OK;
}
extern "C" int suma_map_download(suma_ctx* c, suma_surfel* host, uint32_t cap, uint32_t* n) {
  if (!c) return SUMA_ERR_INVALID;
  int r = read_state(c);
  if (r) return r;
  uint32_t S = c->h_ds->n_surfels;
  if (n) *n = S;
  uint32_t m = S < cap ? S : cap;
  if (m && host) {
    CK(hipMemcpyAsync(host, c->surfels[c->cur], (size_t)m * sizeof(suma_surfel), hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream));
  }
  return SUMA_OK;
}
extern "C" int suma_map_export_surfels(suma_ctx* c, void** d_ptr, uint32_t* n) {
  if (!c || !d_ptr || !n) return SUMA_ERR_INVALID;
  int r = read_state(c);
  if (r) return r;
  *d_ptr = (void*)c->surfels[c->cur];
  *n = c->h_ds->n_surfels;
  return check_overflow(c);
}
extern "C" int suma_map_export_data_surfels(suma_ctx* c, void** d_ptr, uint32_t* first, uint32_t* n_data) {
  if (!c || !d_ptr || !first || !n_data) return SUMA_ERR_INVALID;
  int r = read_state(c);
  if (r) return r;
  *d_ptr = (void*)c->surfels[c->cur];
  *first = c->h_ds->n_kept_updated;
  *n_data = c->h_ds->n_kept_data;
  return check_overflow(c);
}
extern "C" int suma_map_upload(suma_ctx* c, const suma_surfel* host, uint32_t n, uint32_t timestamp) {
  if (!c || (!host && n)) return SUMA_ERR_INVALID;
  if (n > c->p.max_surfels) n = c->p.max_surfels;
  if (n) CK(hipMemcpyAsync(c->surfels[c->cur], host, (size_t)n * sizeof(suma_surfel), hipMemcpyHostToDevice, c->stream));
  CK(hipMemcpyAsync(&c->ds->n_surfels, &n, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  CK(hipStreamSynchronize(c->stream));
  c->timestamp = timestamp;
  c->known_surfels = n;
  c->map_version++;
  return SUMA_OK;
}
extern "C" int suma_map_download_index_map(suma_ctx* c, uint32_t* host) {
  if (!c || !host) return SUMA_ERR_INVALID;
  CK(hipMemcpyAsync(host, c->index_map, c->P * 4, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return SUMA_OK;
}
extern "C" int suma_map_download_radius_conf(suma_ctx* c, suma_float4* host) {
  if (!c || !host) return SUMA_ERR_INVALID;
  CK(hipMemcpyAsync(host, c->radius_conf, c->P * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return SUMA_OK;
}
/* SurfelMap::poses_ (SurfelMap.h:205-208): the pose table the surfels refer to by creation stamp, entries 0 .. timestamp - 1,
 * column-major floats -- the trajectory as the map holds it (after updatePoses: the optimised one) */
extern "C" int suma_map_download_poses(suma_ctx* c, float* host, uint32_t capacity, uint32_t* n) {
  if (!c || !n || (capacity && !host)) return SUMA_ERR_INVALID;
  *n = c->timestamp;
  const uint32_t m = c->timestamp < capacity ? c->timestamp : capacity;
  if (m) {
    CK(hipMemcpyAsync(host, c->poses, (size_t)m * 16 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream));
    host_synced(c);
  }
  return SUMA_OK;
}
extern "C" int suma_map_download_integrated(suma_ctx* c, uint8_t* host) {
  if (!c || !host) return SUMA_ERR_INVALID;
  CK(hipMemcpyAsync(host, c->integrated, c->P, hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  return SUMA_OK;
}
extern "C" int suma_map_counts(suma_ctx* c, uint32_t* n_updated, uint32_t* n_new, uint32_t* n_cached,
                               int32_t origin_ij[2]) {
  if (!c) return SUMA_ERR_INVALID;
  int r = read_state(c);
  if (r) return r;
  if (n_updated) *n_updated = c->h_ds->n_updated;
  if (n_new) *n_new = c->h_ds->n_data;
  if (n_cached) {
    uint32_t ns = (uint32_t)c->cache_index.size();
    std::vector<CacheSlot> slots(ns);
    if (ns) {
      CK(hipMemcpyAsync(slots.data(), c->cache_slots, ns * sizeof(CacheSlot), hipMemcpyDeviceToHost, c->stream));
      CK(hipStreamSynchronize(c->stream));
    }
    uint32_t s = 0;
    for (auto& q : slots) s += q.count;
    *n_cached = s;
  }
  if (origin_ij) {
    origin_ij[0] = c->origin_i;
    origin_ij[1] = c->origin_j;
  }
  return SUMA_OK;
}

extern "C" int suma_map_download_cached_tile(suma_ctx* c, int32_t i, int32_t j, suma_surfel* host, uint32_t capacity,
                                             uint32_t* n) {
  if (!c || !n || (capacity && !host)) return SUMA_ERR_INVALID;
  *n = 0;
  auto it = c->cache_index.find(std::make_pair(i, j));
  if (it == c->cache_index.end()) return SUMA_OK;
  CacheSlot q;
  CK(hipMemcpyAsync(&q, c->cache_slots + it->second, sizeof(CacheSlot), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  *n = q.count;
  const uint32_t m = q.count < capacity ? q.count : capacity;
  if (m) {
    CK(hipMemcpyAsync(host, c->cache_arena + q.offset, (size_t)m * sizeof(suma_surfel), hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream));
  }
  return check_overflow(c);
}

extern "C" int suma_map_cache_stats(suma_ctx* c, uint32_t* used, uint32_t* capacity, uint32_t* compactions) {
  if (!c) return SUMA_ERR_INVALID;
  int r = read_state(c);
  if (r) return r;
  if (used) *used = c->h_ds->cache_used;
  if (capacity) *capacity = c->cache_cap;
  if (compactions) *compactions = c->cache_compactions;
  return SUMA_OK;
}

/* ---------------------------------------------------------------------------------------------
 * loop-closure verification (SurfelMapping.cpp:662-757)
 * ------------------------------------------------------------------------------------------- */
static void mul4_dd(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] =
          ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}

/* One speculative round of the verification: the chains of `n` initial guesses as ONE batched minimisation (grid.y =
 * guess) against whatever model frame the objective points at, then -- still without a host round trip -- the
 * jacobianProducts evaluation at the pose each chain ended on (SurfelMapping.cpp:705), as one more batched pixel pass
 * with eval_only set on the SAME chain states (the state's Frame2Model::iteration_ has kept running, exactly the value
 * the sequential form passes in) and its consume launch.  One copy + one synchronisation for all guesses. */
static int verify_round(suma_ctx* c, const double* inits, uint32_t n, uint32_t iteration0) {
  /* Frame2Model::iteration_ is reset by setData only (Frame2Model.cpp:117-123) and counts on across the guesses
   * (SurfelMapping.cpp:693-700): the first guess of a round starts where the caller says, every later one behind at
   * least one increment of its predecessor.  The shader reads the counter as `iteration > 0` and nothing else
   * (Frame2Model_jacobians.geom:129, the Tukey weight), so "1" stands for the count a speculative chain cannot know. */
  int r = enqueue_minimize(c, inits, n, 0, iteration0, iteration0 > 0 ? iteration0 : 1u); /* :700, n chains side by side */
  if (r) return r;
  const uint32_t iter_arg = c->p.max_iterations > 0 ? c->p.max_iterations : 0xffffffffu;
  {
    ProfScope ps(c, "k6_icp_step", 96.0 * (double)c->icp_current->width * c->icp_current->height * n);
    CK(launch_icp_iteration(c, n, iter_arg, (double)c->p.stopping_threshold, (double)c->p.delta, 1, 0, 1)); /* :705 */
  }
  {
    ProfScope ps(c, "k6_icp_finish", 0.0);
    CK(launch_icp_iteration(c, n, iter_arg, (double)c->p.stopping_threshold, (double)c->p.delta, 1, 0, 0));
  }
  CK(hipMemcpyAsync(c->h_gn, gn_result(c), (size_t)n * sizeof(GnState), hipMemcpyDeviceToHost, c->stream));
  CK(hipStreamSynchronize(c->stream));
  host_synced(c);
  return SUMA_OK;
}

/* SurfelMapping.cpp:679-757 with the initial guesses BATCHED.  The reference minimises the guesses
 * one after the other against oldMapFrame(); its quirk: the first guess that passes the gates re-points the objective at
 * composedFrame() (:718-719) and every LATER guess is minimised against that frame.  So the batch is speculative: all
 * remaining guesses run side by side against the current model; the host walks the results in order and, at the first
 * one that passes, renders the composed view, evaluates it (:720-723) and throws the later speculative results away --
 * they are redone, again as one batch, against the composed frame.  No guess passes (the common case: a candidate is
 * rejected) or only the last one does: ONE chain of launches and one synchronisation for the whole verification, where
 * the sequential form pays n_init chains and 2 n_init host round trips.  Results are the sequential form's bit for bit
 * (suma_loop_closure_verify_serial, kept for the cross-check and the A/B timing). */
extern "C" int suma_loop_closure_verify(suma_ctx* c, const suma_frame* current, const double pose_prior[16],
                                        const double* initializations, uint32_t n_init, const float pose_new[16],
                                        float conf_threshold, float min_valid_ratio, float max_outlier_ratio,
                                        suma_loop_result* out) {
  if (!c || !current || !pose_prior || !initializations || !pose_new || !out) return SUMA_ERR_INVALID;
  float prior_f[16];
  for (int i = 0; i < 16; ++i) prior_f[i] = (float)pose_prior[i];
  int r = suma_map_render_inactive(c, prior_f, conf_threshold); /* :679 */
  if (r) return r;
  r = suma_icp_set_data(c, current, c->old_frame); /* :693 */
  if (r) return r;
  uint32_t start = 0;
  uint32_t iteration0 = 0; /* Frame2Model::iteration_ in front of guess `start`: 0 behind a setData (:693, :719) */
  while (start < n_init) {
    const uint32_t n = (n_init - start) < SUMA_MAX_HYP ? (n_init - start) : SUMA_MAX_HYP;
    r = verify_round(c, initializations + 16 * (size_t)start, n, iteration0);
    if (r) return r;
    iteration0 = 1; /* a round that ends without a pass (more guesses than SUMA_MAX_HYP): the counter has moved */
    uint32_t next = start + n;
    for (uint32_t k = start; k < start + n; ++k) {
      const GnState& g = c->h_gn[k - start];
      suma_loop_result* o = &out[k];
      memset(o, 0, sizeof(*o));
      memcpy(o->gn_pose, g.Tk, sizeof(g.Tk));
      fill_stats(g, &o->after_minimize);
      const suma_icp_stats& s0 = o->after_minimize;
      const float valid_ratio = (float)s0.valid / (float)(s0.valid + s0.invalid);
      const float outlier_ratio = (float)s0.outlier / (float)(s0.outlier + s0.inlier);
      double pd[16];
      mul4_dd(pose_prior, o->gn_pose, pd);
      for (int i = 0; i < 16; ++i) o->pose_old[i] = (float)pd[i];
      o->passed = (valid_ratio > min_valid_ratio && outlier_ratio < max_outlier_ratio) ? 1 : 0; /* :713 */
      if (o->passed) {
        r = suma_map_render_composed(c, o->pose_old, pose_new, conf_threshold); /* :717 */
        if (r) return r;
        r = suma_icp_set_data(c, current, c->composed_frame); /* :719 -- the model of every later guess */
        if (r) return r;
        double I[16];
        for (int i = 0; i < 16; ++i) I[i] = (i % 5 == 0) ? 1.0 : 0.0;
        r = suma_icp_jacobian_products(c, I, 0, o->JtJ, nullptr, nullptr, &o->composed); /* :720-723; overwrites h_gn[0] */
        if (r) return r;
        next = k + 1; /* what was speculated behind this guess saw the wrong model: redo it */
        iteration0 = 0; /* setData at :719; the evaluation at identity increments nothing */
        break;
      }
    }
    start = next;
  }
  return SUMA_OK;
}

/* the reference's sequencing literally: one minimisation, one evaluation and two host round trips per guess */
extern "C" int suma_loop_closure_verify_serial(suma_ctx* c, const suma_frame* current, const double pose_prior[16],
                                        const double* initializations, uint32_t n_init, const float pose_new[16],
                                        float conf_threshold, float min_valid_ratio, float max_outlier_ratio,
                                        suma_loop_result* out) {
  if (!c || !current || !pose_prior || !initializations || !pose_new || !out) return SUMA_ERR_INVALID;
  float prior_f[16];
  for (int i = 0; i < 16; ++i) prior_f[i] = (float)pose_prior[i];
  int r = suma_map_render_inactive(c, prior_f, conf_threshold); /* :679 */
  if (r) return r;
  r = suma_icp_set_data(c, current, c->old_frame); /* :693 */
  if (r) return r;
  uint32_t iteration = 0; /* Frame2Model::iteration_: reset by setData (:693, :719) only, it counts on across the guesses */
  for (uint32_t k = 0; k < n_init; ++k) {
    suma_loop_result* o = &out[k];
    memset(o, 0, sizeof(*o));
    suma_icp_stats mst;
    c->icp_iteration0 = iteration;
    r = suma_icp_minimize(c, initializations + 16 * (size_t)k, o->gn_pose, nullptr, 0, nullptr, &mst); /* :700 */
    if (r) return r;
    /* objective_->jacobianProducts(JtJ, Jtr) at the pose the minimisation left (:705); one increment per step, also
     * the converged one (Objective.h:45-48) -- the counter only matters for the Tukey weight */
    iteration += mst.iterations + (mst.converged ? 1u : 0u);
    r = suma_icp_jacobian_products(c, o->gn_pose, iteration, nullptr, nullptr, nullptr, &o->after_minimize);
    if (r) return r;
    o->after_minimize.iterations = mst.iterations;
    o->after_minimize.converged = mst.converged;
    const suma_icp_stats& s0 = o->after_minimize;
    const float valid_ratio = (float)s0.valid / (float)(s0.valid + s0.invalid);
    const float outlier_ratio = (float)s0.outlier / (float)(s0.outlier + s0.inlier);
    double pd[16];
    mul4_dd(pose_prior, o->gn_pose, pd);
    for (int i = 0; i < 16; ++i) o->pose_old[i] = (float)pd[i];
    o->passed = (valid_ratio > min_valid_ratio && outlier_ratio < max_outlier_ratio) ? 1 : 0; /* :713 */
    if (o->passed) {
      r = suma_map_render_composed(c, o->pose_old, pose_new, conf_threshold); /* :717 */
      if (r) return r;
      r = suma_icp_set_data(c, current, c->composed_frame); /* :719 -- stays set for the next guess */
      if (r) return r;
      double I[16];
      for (int i = 0; i < 16; ++i) I[i] = (i % 5 == 0) ? 1.0 : 0.0;
      r = suma_icp_jacobian_products(c, I, 0, o->JtJ, nullptr, nullptr, &o->composed); /* :720-723 */
      if (r) return r;
      iteration = 0; /* setData at :719 */
    }
  }
  return SUMA_OK;
}

/* SE3::log, lie_algebra.cpp:36-71 (host side, double, libm) */
extern "C" void suma_se3_log(const double T[16], double x[6]) {
  /* column-major: R(r, c) = T[4 * c + r] */
  const double d = 0.5 * (((T[0] + T[5]) + T[10]) - 1.0);
  double W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* omega_skew, row-major W[3 * r + c] */
  for (int i = 0; i < 6; ++i) x[i] = 0.0;
  if (d < 1 - 1e-10) {
    const double theta = acos(d);
    const double f = theta / (2 * sin(theta));
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) W[3 * r + cc] = f * (T[4 * cc + r] - T[4 * r + cc]);
    x[3] = W[3 * 2 + 1];
    x[4] = W[3 * 0 + 2];
    x[5] = W[3 * 1 + 0];
  }
  const double theta = sqrt((x[3] * x[3] + x[4] * x[4]) + x[5] * x[5]);
  const double t[3] = {T[12], T[13], T[14]};
  x[0] = t[0];
  x[1] = t[1];
  x[2] = t[2];
  if (fabs(theta) > 1e-10) {
    const double half_theta = 0.5 * theta;
    const double alpha = -0.5;
    const double beta = 1 / (theta * theta) * (1 - theta * cos(half_theta) / (2 * sin(half_theta)));
    double W2[9], Vi[9];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc)
        W2[3 * r + cc] = (W[3 * r] * W[cc] + W[3 * r + 1] * W[3 + cc]) + W[3 * r + 2] * W[6 + cc];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) Vi[3 * r + cc] = ((r == cc ? 1.0 : 0.0) + alpha * W[3 * r + cc]) + beta * W2[3 * r + cc];
    for (int r = 0; r < 3; ++r) x[r] = (Vi[3 * r] * t[0] + Vi[3 * r + 1] * t[1]) + Vi[3 * r + 2] * t[2];
  }
}

/* checkLoopClosure, part 1 (SurfelMapping.cpp:546-574): a closure that is being tracked is verified again */
extern "C" int suma_loop_closure_track(suma_ctx* c, const suma_frame* current, const double last_pose_old[16],
                                       const double last_increment[16], const float pose_new[16], float conf_threshold,
                                       double min_valid_ratio, double max_outlier_ratio, double max_increment_difference,
                                       suma_loop_track* o) {
  if (!c || !current || !last_pose_old || !last_increment || !pose_new || !o) return SUMA_ERR_INVALID;
  memset(o, 0, sizeof(*o));
  float pf[16];
  for (int i = 0; i < 16; ++i) pf[i] = (float)last_pose_old[i]; /* :548 */
  int r = suma_map_render_inactive(c, pf, conf_threshold); /* :550 */
  if (r) return r;
  r = suma_icp_set_data(c, current, c->old_frame); /* :553 */
  if (r) return r;
  r = suma_icp_minimize(c, last_increment, o->increment_old, nullptr, 0, nullptr, &o->after_minimize); /* :554 */
  if (r) return r;
  const suma_icp_stats& s0 = o->after_minimize; /* the objective's counters as the last step left them (:557-558) */
  const float valid_ratio = (float)s0.valid / (float)(s0.valid + s0.invalid);
  const float outlier_ratio = (float)s0.outlier / (float)(s0.outlier + s0.inlier);
  double la[6], lb[6], sq = 0.0;
  suma_se3_log(last_increment, la);
  suma_se3_log(o->increment_old, lb);
  for (int i = 0; i < 6; ++i) sq += (la[i] - lb[i]) * (la[i] - lb[i]);
  o->increment_difference = (float)sqrt(sq); /* :561 */
  mul4_dd(last_pose_old, o->increment_old, o->pose_old);
  o->passed = ((double)valid_ratio > min_valid_ratio && (double)outlier_ratio < max_outlier_ratio &&
               (double)o->increment_difference < max_increment_difference) ? 1 : 0; /* :563: floats against double literals */
  if (o->passed) {
    float po[16];
    for (int i = 0; i < 16; ++i) po[i] = (float)o->pose_old[i]; /* :564 */
    r = suma_map_render_composed(c, po, pose_new, conf_threshold); /* :567 */
    if (r) return r;
    r = suma_icp_set_data(c, current, c->composed_frame); /* :569 */
    if (r) return r;
    double I[16];
    for (int i = 0; i < 16; ++i) I[i] = (i % 5 == 0) ? 1.0 : 0.0;
    r = suma_icp_jacobian_products(c, I, 0, o->JtJ, nullptr, nullptr, &o->composed); /* :570-572 */
    if (r) return r;
  }
  return SUMA_OK;
}

/* ---------------------------------------------------------------------------------------------
 * SurfelMapping::processScan
 * ------------------------------------------------------------------------------------------- */
static int resolve_stats(suma_pipeline* s, bool need_sync);

static void eye_d(double* T) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
static void mul4_d(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] =
          ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}
static void rigid_inv_d(const double* m, double* out) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) out[4 * c + r] = m[4 * r + c];
  for (int r = 0; r < 3; ++r) out[12 + r] = -((m[4 * r] * m[12] + m[4 * r + 1] * m[13]) + m[4 * r + 2] * m[14]);
  out[3] = out[7] = out[11] = 0.0;
  out[15] = 1.0;
}
static void cast_f(const double* T, float* out) {
  for (int i = 0; i < 16; ++i) out[i] = (float)T[i];
}

extern "C" int suma_pipeline_create(const suma_params* params, int hip_device, suma_pipeline** out) {
  if (!params || !out) return SUMA_ERR_INVALID;
  *out = nullptr;
  suma_ctx* c = nullptr;
  int r = suma_ctx_create(params, hip_device, &c);
  if (r) return r;
  suma_pipeline* s = new (std::nothrow) suma_pipeline();
  if (!s) {
    suma_ctx_destroy(c);
    return SUMA_ERR_NOMEM;
  }
  memset(s, 0, sizeof(*s));
  s->c = c;
  suma_frame** fr[4] = {&s->last_frame, &s->current_frame, &s->current_model, &s->last_model};
  for (int k = 0; k < 4; ++k) {
    bool data = k < 2;
    r = frame_create_raw(c, data ? params->data_width : params->model_width,
                         data ? params->data_height : params->model_height, fr[k]);
    if (r) {
      g_create_error = c->err;
      suma_pipeline_destroy(s);
      return r;
    }
  }
  if (hipHostMalloc((void**)&s->h_res, 3 * sizeof(HostResult), hipHostMallocDefault) != hipSuccess) {
    g_create_error = "hipHostMalloc failed";
    suma_pipeline_destroy(s);
    return SUMA_ERR_HIP;
  }
  memset(s->h_res, 0, 3 * sizeof(HostResult));
  /* side stream for work off the critical path of a scan (k_sync.hip); SUMA_NO_SIDE_STREAM=1 or a tool that
   * serialises kernel execution keeps everything on the ctx stream (ensure_side_stream) */
  if (ensure_side_stream(c) != SUMA_OK) {
    g_create_error = "side stream setup failed: " + c->err;
    suma_pipeline_destroy(s);
    return SUMA_ERR_HIP;
  }
  s->res_seq = 0;
  s->stats_pending = false;
  s->stats_slot = 0;
  eye_d(s->current_pose);
  eye_d(s->last_pose);
  eye_d(s->pose_old);
  eye_d(s->pose_new);
  eye_d(s->last_increment);
  eye_d(s->last_pose_old);
  s->phase = 0;
  float p_unstable = 0.1f; /* SurfelMapping.cpp:108-109 */
  s->log_unstable = (float)log((double)(p_unstable / (1.0f - p_unstable)));
  *out = s;
  return SUMA_OK;
}
extern "C" void suma_pipeline_destroy(suma_pipeline* s) {
  if (!s) return;
  ingest_destroy(s->c); /* its threads use the frames and streams freed below */
  if (s->c && s->c->stream) hipStreamSynchronize(s->c->stream);
  suma_frame_destroy(s->last_frame);
  suma_frame_destroy(s->current_frame);
  suma_frame_destroy(s->current_model);
  suma_frame_destroy(s->last_model);
  if (s->h_res) hipHostFree(s->h_res);
  suma_ctx_destroy(s->c);
  delete s;
}
extern "C" suma_ctx* suma_pipeline_ctx(suma_pipeline* s) { return s ? s->c : nullptr; }
extern "C" int suma_pipeline_pose(const suma_pipeline* s, double pose[16]) {
  if (!s || !pose) return SUMA_ERR_INVALID;
  memcpy(pose, s->current_pose, 16 * sizeof(double));
  return SUMA_OK;
}
extern "C" int suma_pipeline_last_increment(const suma_pipeline* s, double inc[16]) {
  if (!s || !inc) return SUMA_ERR_INVALID;
  memcpy(inc, s->last_increment, 16 * sizeof(double));
  return SUMA_OK;
}
extern "C" int suma_pipeline_last_stats(const suma_pipeline* s, suma_icp_stats* st) {
  if (!s || !st) return SUMA_ERR_INVALID;
  int r = resolve_stats(const_cast<suma_pipeline*>(s), true);
  *st = s->stats;
  return r;
}
extern "C" int suma_pipeline_minimize_stats(const suma_pipeline* s, suma_icp_stats* st) {
  if (!s || !st) return SUMA_ERR_INVALID;
  *st = s->stats_mst;
  return SUMA_OK;
}
extern "C" uint32_t suma_pipeline_timestamp(const suma_pipeline* s) { return s ? s->timestamp : 0; }
extern "C" uint32_t suma_pipeline_track_loss(const suma_pipeline* s) { return s ? s->track_loss : 0; }
extern "C" suma_frame* suma_pipeline_frame(suma_pipeline* s, int which) {
  if (!s) return nullptr;
  return which == 0 ? s->current_frame : (which == 1 ? s->last_model : s->current_model);
}

/* the stream has been synchronised (or will be here): turn the pending statistics record into s->stats.  A
 * record that was never stamped (the statistics launch faulted or was lost) is an error, not stale numbers. */
static int resolve_stats(suma_pipeline* s, bool need_sync) {
  if (!s->stats_pending) return SUMA_OK;
  suma_ctx* c = s->c;
  if (need_sync) CK(hipStreamSynchronize(c->stream));
  /* written by a launch that precedes, in stream order, either the synchronisation above or the
   * minimisation result the caller has just received */
  int r = wait_host_result(c, &s->h_res[1 + s->stats_slot], s->stats_seq);
  s->stats_pending = false;
  if (r) return fail(c, SUMA_ERR_HIP, "statistics pass did not report (launch failed?)");
  suma_icp_stats st;
  fill_stats_host(s->h_res[1 + s->stats_slot], &st);
  st.iterations = s->stats_mst.iterations;
  st.converged = s->stats_mst.converged;
  s->stats = st;
  return SUMA_OK;
}

/* SurfelMapping::getConfidenceThreshold, SurfelMapping.cpp:333-340 (time_init = 10) */
static float conf_threshold(const suma_pipeline* s) {
  float ct = s->c->p.confidence_threshold;
  const uint32_t time_init = 10;
  if (s->timestamp < time_init) {
    float alpha = (float)s->timestamp / (float)time_init;
    ct = (float)((1.0 - (double)alpha) * (double)s->log_unstable + (double)(alpha * s->c->p.confidence_threshold));
  }
  return ct;
}

/* one minimisation with the optional fixed-iteration override; reads back pose + stats + counters */
static int minimize_cfg(suma_pipeline* s, const suma_frame* cur, const suma_frame* model, const double* T0, double* T,
                        int32_t fixed_iterations, suma_icp_stats* st) {
  suma_ctx* c = s->c;
  suma_params saved = c->p;
  if (fixed_iterations > 0) {
    c->p.max_iterations = (uint32_t)fixed_iterations;
    c->p.stopping_threshold = 0.0f;
    c->p.delta = 0.0f;
  }
  c->icp_current = cur;
  c->icp_model = model;
  int r = enqueue_minimize(c, T0, 1, 0);
  c->p = saved;
  if (r) return r;
  CK(hipMemcpyAsync(c->h_gn, gn_result(c), sizeof(GnState), hipMemcpyDeviceToHost, c->stream));
  CK(hipMemcpyAsync(c->h_ds, c->ds, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
  for (;;) { /* poll instead of a blocking wait */
    hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) CK(q);
  }
  int rs = resolve_stats(s, false); /* everything enqueued before this point has completed */
  if (rs) return rs;
  c->known_surfels = c->h_ds->n_surfels;
  memcpy(T, c->h_gn[0].Tk, 16 * sizeof(double));
  fill_stats(c->h_gn[0], st);
  return check_overflow(c);
}

/* SurfelMapping::updatePose, SurfelMapping.cpp:372-476.
 * Everything up to and including the statistics pass depends only on the FIRST minimisation (the
 * reference re-renders and takes its statistics before it looks at the fallback condition,
 * :406-423 vs :434-449), so it is enqueued in one go: the closing Gauss-Newton launch leaves the
 * re-render pose in HBM, and the host reads the increment back while the GPU is already busy with
 * the re-render and the statistics pass -- the stream never drains inside a scan. */
static int update_pose(suma_pipeline* s, int32_t fixed_iterations) {
  suma_ctx* c = s->c;
  double T0[16], increment[16];
  if (!c->p.initialize_identity)
    memcpy(T0, s->last_increment, sizeof(T0));
  else
    eye_d(T0);
  suma_icp_stats mst;
  /* --- frame-to-model minimisation (:384-396), result copied to the host, event recorded --- */
  {
    suma_params saved = c->p;
    if (fixed_iterations > 0) {
      c->p.max_iterations = (uint32_t)fixed_iterations;
      c->p.stopping_threshold = 0.0f;
      c->p.delta = 0.0f;
    }
    c->icp_current = s->current_frame;
    c->icp_model = c->new_frame;
    c->gn_emit_pose = 1;
    memcpy(c->gn_pose_base, s->pose_new, sizeof(c->gn_pose_base));
    s->res_seq += 1;
    c->gn_host_out = &s->h_res[0]; /* the closing launch reports straight into pinned host memory */
    c->gn_host_seq = s->res_seq;
    int r0 = enqueue_minimize(c, T0, 1, 0);
    c->gn_emit_pose = 0;
    c->gn_host_out = nullptr;
    c->p = saved;
    if (r0) return r0;
  }
  /* --- re-render from pose_new * increment (pose taken from HBM), K7 splat and the
   *     lastModelFrame copy fused in (:406-407) --- */
  c->rendered.valid = false;
  CK(launch_map_render_single(c, nullptr, conf_threshold(s), 1, 1, s->last_model));
  c->new_frame->version++;
  s->last_model->version++;
  /* --- statistics pass (:411-423) --- */
  double I[16];
  eye_d(I);
  c->icp_current = s->current_frame;
  c->icp_model = c->new_frame;
  CK(launch_gn_init(c, I, 1, 0, 0));
  const uint32_t slot = s->stats_slot ^ 1u; /* the previous scan's record may not have been looked at yet */
  {
    /* one launch: the pass closes itself (last block totals and reports to the host record) */
    ProfScope ps(c, "k6k8_stats_radius", (96.0 + 81.0) * (double)c->P); /* K6 reads + K8's 81 B of products per pixel */
    c->gn_fused_report = &s->h_res[1 + slot];
    c->gn_host_seq = s->res_seq;
    /* this launch streams the three maps of the current frame: K8's per-pixel work for the update that
     * follows (pose independent) and the per-update counter resets ride along -- no k8_radius launch */
    c->gn_fuse_k8 = 1;
    hipError_t e = launch_icp_iteration(c, 1, 1, 0.0, 0.0, 1, 0, 1);
    c->gn_fuse_k8 = 0;
    c->gn_fused_report = nullptr;
    CK(e);
    c->k8_fused_frame = s->current_frame;
    c->k8_fused_version = s->current_frame->version;
    c->k8_fused_stamp = c->timestamp;
    c->k8_fused_params = c->params_version;
  }
  /* --- wait for the minimisation result only (poll on the record's sequence number) --- */
  struct timespec tw0, tw1;
  clock_gettime(CLOCK_MONOTONIC, &tw0);
  int r = wait_host_result(c, &s->h_res[0], s->res_seq);
  clock_gettime(CLOCK_MONOTONIC, &tw1);
  c->het.result_wait_s += (double)(tw1.tv_sec - tw0.tv_sec) + 1e-9 * (double)(tw1.tv_nsec - tw0.tv_nsec);
  if (r) return r;
  r = resolve_stats(s, false); /* the previous scan's statistics launch precedes this result in the stream */
  if (r) return r;
  s->stats_slot = slot;
  *c->h_ds = s->h_res[0].ds;
  c->known_surfels = c->h_ds->n_surfels;
  memcpy(increment, s->h_res[0].Tk, sizeof(increment));
  fill_stats_host(s->h_res[0], &mst);
  r = check_overflow(c);
  if (r) return r;
  s->stats_mst = mst;
  s->stats_seq = s->res_seq;
  s->stats_pending = true;

  double inv_last[16], delta[16], posed[16];
  float posef[16];
  rigid_inv_d(s->last_increment, inv_last);
  mul4_d(inv_last, increment, delta);
  mul4_d(s->pose_new, increment, posed);
  cast_f(posed, posef); /* the same value the closing launch wrote to HBM */
  c->k7.valid = true;
  c->k7.map_version = c->map_version;
  c->k7.params_version = c->params_version;
  memcpy(c->k7.pose, posef, sizeof(posef));

  float t_err = (float)sqrt((delta[12] * delta[12] + delta[13] * delta[13]) + delta[14] * delta[14]);
  float angle = (float)(0.5 * (((delta[0] + delta[5]) + delta[10]) - 1.0));
  float r_err = (float)acos((double)fmaxf(fminf(angle, 1.0f), -1.0f));
  const bool fallback = (s->timestamp > 1 && ((double)t_err > 0.4 || (double)r_err > 0.1) && c->p.fallback_mode); /* :438-449: float against the double literals */
  if (fallback) {
    s->track_loss += 1;
    suma_params saved = c->p;
    c->p.icp_max_distance = c->p.fallback_max_distance;
    c->p.icp_max_angle = c->p.fallback_max_angle;
    r = minimize_cfg(s, s->current_frame, s->last_frame, T0, increment, fixed_iterations, &mst);
    c->p = saved;
    if (r) return r;
  }
  memcpy(s->last_pose, s->current_pose, sizeof(s->last_pose));
  double np[16];
  mul4_d(s->current_pose, increment, np);
  memcpy(s->current_pose, np, sizeof(np));
  memcpy(s->last_pose_old, s->pose_old, sizeof(s->last_pose_old)); /* :456 */
  memcpy(s->pose_old, np, sizeof(np));
  memcpy(s->pose_new, np, sizeof(np));
  memcpy(s->last_increment, increment, sizeof(increment));
  return SUMA_OK;
}

/* ---- the three phases of SurfelMapping::processScan (SurfelMapping.cpp:175-204) ----
 * upload_done: optional event on another stream that the scan's device buffers depend on (device-side ingest) */
int pipeline_begin_scan_impl(suma_pipeline* s, const suma_float4* d_points, const float* d_labels, const float* d_probs,
                             uint32_t n, hipEvent_t upload_done) {
  if (!s || (n > 0 && !d_points)) return SUMA_ERR_INVALID;
  suma_ctx* c = s->c;
  if (s->phase != 0) return fail(c, SUMA_ERR_INVALID, "suma_pipeline_begin_scan: the previous scan has not been closed with suma_pipeline_update_map");
  c->obj_set = false; /* the pipeline's objective_ runs on the ctx parameters (suma_params), not on a stale adapter object's */
  /* initialize(), SurfelMapping.cpp:323-331 */
  std::swap(s->last_frame, s->current_frame);
  std::swap(s->last_model, s->current_model);
  /* preprocess(), :342-358.  K1-K3 of this scan go to the side stream: the host is ahead of the GPU here (the
   * surfel passes of the previous scan are still running on the ctx stream), so they overlap that tail instead of
   * queueing behind it.  Buffers: the frame written here was last read by work the host has already waited for (the
   * previous scan's minimisation result is behind it in stream order), K1 has its own z-buffer.  The ctx stream
   * continues behind a gate that waits for the side stream's signal (k_sync.hip). */
  int r;
  if (c->side_stream) {
    if (upload_done) HIP_TRY(c, hipStreamWaitEvent(c->side_stream, upload_done, 0));
    c->ls = c->side_stream;
    r = suma_preprocess_device(c, d_points, d_labels, d_probs, n, s->timestamp, s->current_frame);
    c->ls = c->stream;
    if (r) return r;
    r = side_handoff(c, s->current_frame); /* the first reader of the frame on the ctx stream issues the wait (flush_gate) */
    if (r) return r;
  } else {
    if (upload_done) HIP_TRY(c, hipStreamWaitEvent(c->stream, upload_done, 0));
    r = suma_preprocess_device(c, d_points, d_labels, d_probs, n, s->timestamp, s->current_frame);
    if (r) return r;
  }
  float po[16], pn[16];
  cast_f(s->pose_old, po);
  cast_f(s->pose_new, pn);
  r = map_render_dedup(c, po, pn, conf_threshold(s), s->last_model);
  if (r) return r;
  s->phase = 1;
  return SUMA_OK;
}

int pipeline_update_pose_impl(suma_pipeline* s, int32_t fixed_iterations) {
  if (!s) return SUMA_ERR_INVALID;
  if (s->phase != 1) return fail(s->c, SUMA_ERR_INVALID, "suma_pipeline_update_pose: call suma_pipeline_begin_scan first");
  if (s->timestamp > 0) { /* :190 */
    int r = update_pose(s, fixed_iterations);
    if (r) return r;
  }
  s->phase = 2;
  return SUMA_OK;
}

/* updateMap(), :797-804, and timestamp_ += 1 (:209) */
int pipeline_update_map_impl(suma_pipeline* s) {
  if (!s) return SUMA_ERR_INVALID;
  suma_ctx* c = s->c;
  if (s->phase != 2) return fail(c, SUMA_ERR_INVALID, "suma_pipeline_update_map: call suma_pipeline_update_pose first");
  float pc[16];
  cast_f(s->current_pose, pc);
  int r = suma_map_update(c, pc, s->current_frame);
  if (r) return r;
  r = map_render_dedup(c, pc, pc, conf_threshold(s), s->current_model);
  if (r) return r;
  s->timestamp += 1;
  s->phase = 0;
  return SUMA_OK;
}

hipStream_t pipeline_input_stream(suma_pipeline* s) { return s->c->side_stream ? s->c->side_stream : s->c->stream; }

int pipeline_process_scan_impl(suma_pipeline* s, const suma_float4* d_points, const float* d_labels,
                               const float* d_probs, uint32_t n, int32_t fixed_iterations, hipEvent_t upload_done) {
  int r = pipeline_begin_scan_impl(s, d_points, d_labels, d_probs, n, upload_done);
  if (r == SUMA_OK) r = pipeline_update_pose_impl(s, fixed_iterations);
  if (r == SUMA_OK) r = pipeline_update_map_impl(s);
  if (r != SUMA_OK && s) s->phase = 0; /* a failed scan does not wedge the phase check */
  return r;
}

extern "C" int suma_pipeline_begin_scan_device(suma_pipeline* s, const suma_float4* d_points, const float* d_labels,
                                               const float* d_probs, uint32_t n) {
  return pipeline_begin_scan_impl(s, d_points, d_labels, d_probs, n, nullptr);
}
extern "C" int suma_pipeline_update_pose(suma_pipeline* s, int32_t fixed_iterations) {
  return pipeline_update_pose_impl(s, fixed_iterations);
}
extern "C" int suma_pipeline_update_map(suma_pipeline* s) { return pipeline_update_map_impl(s); }

/* SurfelMapping::reset (SurfelMapping.cpp:131-169): empty map, identity poses, timestamp 0 -- the object is as new */
extern "C" int suma_pipeline_reset(suma_pipeline* s) {
  if (!s) return SUMA_ERR_INVALID;
  suma_ctx* c = s->c;
  int r = suma_synchronize(c);
  if (r) return r;
  if (c->gate_pending) c->gate_pending = 0; /* both streams have drained */
  host_synced(c);
  ingest_drain(c); /* scans staged ahead by a sequence that ended early are not this pipeline's next scans */
  r = map_reset_impl(c); /* also clears an index-map splat nobody will consume */
  if (r) return r;
  s->stats_pending = false;
  memset(&s->stats, 0, sizeof(s->stats));
  s->timestamp = 0;
  s->track_loss = 0;
  s->phase = 0;
  eye_d(s->current_pose);
  eye_d(s->last_pose);
  eye_d(s->pose_old);
  eye_d(s->pose_new);
  eye_d(s->last_increment);
  eye_d(s->last_pose_old);
  c->obj_set = false;
  c->k8_fused_frame = nullptr;
  return suma_synchronize(c);
}

/* ---- hypothesis tracking on the scan pipeline (BASELINE config 3; the reference's pattern of several minimisations of
 *      one frame pair from different starts, SurfelMapping.cpp:662-779, applied to odometry): between begin_scan and
 *      update_map, INSTEAD of update_pose -- the caller minimises a batch of starts against the rendered model
 *      (map_->newMapFrame(), as updatePose does, :384), picks one result and applies it. */
extern "C" int suma_pipeline_minimize_hypotheses(suma_pipeline* s, const double* T0s, uint32_t n_hyp,
                                                 int32_t fixed_iterations, double* T_out, suma_icp_stats* stats) {
  if (!s || !T0s || !T_out || n_hyp == 0 || n_hyp > SUMA_MAX_HYP) return SUMA_ERR_INVALID;
  suma_ctx* c = s->c;
  if (s->phase != 1) return fail(c, SUMA_ERR_INVALID, "suma_pipeline_minimize_hypotheses: between suma_pipeline_begin_scan and suma_pipeline_apply_increment");
  suma_params saved = c->p;
  if (fixed_iterations > 0) {
    c->p.max_iterations = (uint32_t)fixed_iterations;
    c->p.stopping_threshold = 0.0f;
    c->p.delta = 0.0f;
  }
  c->icp_current = s->current_frame;
  c->icp_model = c->new_frame;
  /* this path never runs update_pose, whose result record carries DevState: fetch it with the batch (the copy rides in
   * front of the batch's own read-back and synchronisation), so that the map size the grids are sized from and the
   * capacity / arena / time-out bits reach the host on every scan (round-3 advisor) */
  hipError_t e = hipMemcpyAsync(c->h_ds, c->ds, sizeof(DevState), hipMemcpyDeviceToHost, c->stream);
  int r = (e == hipSuccess) ? suma_icp_minimize_batch(c, T0s, n_hyp, T_out, stats) : SUMA_ERR_HIP;
  c->p = saved;
  if (e != hipSuccess) c->err = std::string("hipMemcpyAsync(DevState): ") + hipGetErrorString(e);
  if (r) return r;
  c->known_surfels = c->h_ds->n_surfels;
  return check_overflow(c);
}
/* the pose bookkeeping of updatePose (SurfelMapping.cpp:453-474) for an increment chosen by the caller */
extern "C" int suma_pipeline_apply_increment(suma_pipeline* s, const double increment[16]) {
  if (!s || !increment) return SUMA_ERR_INVALID;
  if (s->phase != 1) return fail(s->c, SUMA_ERR_INVALID, "suma_pipeline_apply_increment: call suma_pipeline_begin_scan first");
  memcpy(s->last_pose, s->current_pose, sizeof(s->last_pose));
  double np[16];
  mul4_d(s->current_pose, increment, np);
  memcpy(s->current_pose, np, sizeof(np));
  memcpy(s->last_pose_old, s->pose_old, sizeof(s->last_pose_old));
  memcpy(s->pose_old, np, sizeof(np));
  memcpy(s->pose_new, np, sizeof(np));
  memcpy(s->last_increment, increment, 16 * sizeof(double));
  s->phase = 2;
  return SUMA_OK;
}

/* integrateLoopClosures, SurfelMapping.cpp:211-250 (the part behind the optimiser's future) */
extern "C" int suma_pipeline_integrate_loop_closures(suma_pipeline* s, const float* poses16, uint32_t n,
                                                     const double difference[16]) {
  if (!s || !difference || (!poses16 && n)) return SUMA_ERR_INVALID;
  suma_ctx* c = s->c;
  if (s->phase != 0) return fail(c, SUMA_ERR_INVALID, "suma_pipeline_integrate_loop_closures: only between scans (SurfelMapping.cpp:179)");
  int r = suma_map_update_poses(c, poses16, n); /* :236 */
  if (r) return r;
  double np[16];
  mul4_d(difference, s->current_pose, np); /* :239 */
  memcpy(s->current_pose, np, sizeof(np));
  memcpy(s->pose_old, np, sizeof(np)); /* :243 */
  memcpy(s->pose_new, np, sizeof(np));
  return SUMA_OK;
}
extern "C" int suma_pipeline_set_pose_old(suma_pipeline* s, const double pose_old[16]) {
  if (!s || !pose_old) return SUMA_ERR_INVALID;
  memcpy(s->pose_old, pose_old, sizeof(s->pose_old));
  return SUMA_OK;
}
extern "C" int suma_pipeline_get_pose(const suma_pipeline* s, int which, double pose[16]) {
  if (!s || !pose || which < 0 || which > 4) return SUMA_ERR_INVALID;
  const double* src[5] = {s->current_pose, s->pose_old, s->pose_new, s->last_pose_old, s->last_pose};
  memcpy(pose, src[which], 16 * sizeof(double));
  return SUMA_OK;
}
extern "C" int suma_pipeline_result_new(suma_pipeline* s, suma_icp_stats* st) { return suma_pipeline_last_stats(s, st); }

extern "C" int suma_pipeline_verify_loop_closure(suma_pipeline* s, const double pose_prior[16],
                                                 const double* initializations, uint32_t n_init, float min_valid_ratio,
                                                 float max_outlier_ratio, suma_loop_result* out) {
  if (!s) return SUMA_ERR_INVALID;
  if (s->phase != 2) return fail(s->c, SUMA_ERR_INVALID, "suma_pipeline_verify_loop_closure: between suma_pipeline_update_pose and suma_pipeline_update_map (SurfelMapping.cpp:196)");
  float pn[16];
  cast_f(s->pose_new, pn); /* currentPose_new_.cast<float>(), :717 */
  return suma_loop_closure_verify(s->c, s->current_frame, pose_prior, initializations, n_init, pn, conf_threshold(s),
                                  min_valid_ratio, max_outlier_ratio, out);
}
extern "C" int suma_pipeline_track_loop_closure(suma_pipeline* s, double min_valid_ratio, double max_outlier_ratio,
                                                double max_increment_difference, suma_loop_track* out) {
  if (!s) return SUMA_ERR_INVALID;
  if (s->phase != 2) return fail(s->c, SUMA_ERR_INVALID, "suma_pipeline_track_loop_closure: between suma_pipeline_update_pose and suma_pipeline_update_map (SurfelMapping.cpp:196)");
  float pn[16];
  cast_f(s->pose_new, pn);
  return suma_loop_closure_track(s->c, s->current_frame, s->last_pose_old, s->last_increment, pn, conf_threshold(s),
                                 min_valid_ratio, max_outlier_ratio, max_increment_difference, out);
}

extern "C" int suma_pipeline_process_scan_device(suma_pipeline* s, const suma_float4* d_points, const float* d_labels,
                                                 const float* d_probs, uint32_t n, int32_t fixed_iterations) {
  return pipeline_process_scan_impl(s, d_points, d_labels, d_probs, n, fixed_iterations, nullptr);
}

extern "C" int suma_pipeline_process_scan(suma_pipeline* s, const suma_float4* points, const float* labels,
                                          const float* probs, uint32_t n, int32_t fixed_iterations) {
  if (!s || (n > 0 && !points)) return SUMA_ERR_INVALID;
  suma_ctx* c = s->c;
  /* hand-over of a pageable host scan (the reference's glBufferData in Frame::points.assign,
   * Preprocessing.cpp:123-125): copied to pinned memory by several threads, uploaded on the copy stream, consumed
   * behind a device-side dependency (suma_ingest.hip); scans that should be staged AHEAD of their turn go through
   * suma_pipeline_prefetch_scan / process_prefetched */
  (void)c;
  return pipeline_process_host_scan(s, points, labels, probs, n, fixed_iterations);
}
