/*
 * k_update.hip -- K7..K12: SurfelMap::update on gfx950.
 *
 * Replaces (reference):
 *   SurfelMap::update                 src/core/SurfelMap.cpp:492-584
 *   K7  renderIndexmap                SurfelMap.cpp:586-604 + src/shader/gen_indexmap.vert:62-81
 *   K8  generateDataSurfels           SurfelMap.cpp:606-619 + init_radiusConf.vert:34-68
 *   K9  updateSurfels (update)        SurfelMap.cpp:621-644 + update_surfels.vert:140-334,
 *                                     update_surfels.geom:30-43, update_surfels.frag:9-12
 *   K10 updateSurfels (initialise)    SurfelMap.cpp:646-664 + gen_surfels.vert:38-52, gen_surfels.geom:109-145
 *   K11 copySurfels                   SurfelMap.cpp:667-698 + copy_surfels.vert:38-56
 *   K12 extractSurfels                SurfelMap.cpp:708-742 + extract_surfels.vert:46-64
 *
 * GL structure replaced: transform feedback (an order-preserving append performed by the
 * fixed-function pipeline) becomes a single-pass stable compaction: every SUMA_TILE (1024) surfel tile
 * computes its survivors, publishes its count in an 8-byte status word {30-bit launch epoch, flag, count} and in
 * the accumulator of its group of 64 tiles, and obtains its output offset from the preceding groups / tiles
 * (tile ids handed out by an atomic ticket, so predecessors are always running; see lookback_prefix).
 * Output order = input order, exactly as transform feedback guarantees.  K11's area filter is evaluated inside K9 / K10, so the map is
 * read once and written once per scan instead of being copied a second time:
 *   traffic per scan = 64 B * (S + S_new) + the gathered measurement texels.
 * Point splats with depth test (K7, the K9 integration mask) are 64-bit atomicMin / plain
 * stores as in k_preprocess.hip.
 */
#include <cstddef>
#include <cstring>

#include "suma_internal.h"

/* ---------------------------------------------------------------------------------------------
 * decoupled look-back
 * ------------------------------------------------------------------------------------------- */
#define ST_AGG 1u
#define ST_INC 2u

__device__ __forceinline__ unsigned long long st_pack(uint32_t epoch, uint32_t flag, uint32_t value) {
  return ((unsigned long long)epoch << 34) | ((unsigned long long)flag << 32) | value;
}

/* Two-level prefix for the single-pass stable compaction.  Called by wave 0 of a block; returns
 * the number of selected items in all tiles before `tile`.
 *
 * All tiles of a launch run at (nearly) the same time on this chip (2048 resident blocks), so the
 * classic decoupled look-back degenerates into a serial walk over windows of aggregate-only
 * predecessors (one L2/fabric round trip per 64 tiles).  Instead every tile publishes its count
 * twice: as an 8-byte status word {epoch, count} and into the 64-bit accumulator of its GROUP of
 * 64 tiles ({tiles arrived, sum} updated by one atomic add).  A tile then needs
 *      sum over complete groups before its own  +  sum over the earlier tiles of its own group,
 * i.e. ceil(g / 64) + 1 wave-wide loads, all independent and all issued at once; lanes spin only
 * on words that are not published yet.  Critical path = compute + one publish + one read.
 * Status words carry the launch epoch (never cleared); group accumulators are cleared by the
 * finaliser of each launch (block_leaves_last). */
/* Every count travels as a PAIR {items selected, items selected that also go to the submap cache}: a tile's
 * status value is (second << 16) | first (a tile holds at most 4096 items), a group accumulator is
 * {tiles arrived : 8, sum of second : 24, sum of first : 32}.  Kernels with one count pass second = 0. */
struct Pair {
  uint32_t a, x;
};
__device__ __forceinline__ void lookback_publish(unsigned long long* __restrict__ status,
                                                 unsigned long long* __restrict__ group, uint32_t tile, uint32_t agg,
                                                 uint32_t epoch, uint32_t aggx = 0) {
  __hip_atomic_store(&status[tile], st_pack(epoch, ST_AGG, (aggx << 16) | agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(&group[tile >> 6], (1ull << 56) | ((unsigned long long)aggx << 32) | (unsigned long long)agg,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* called by one full wave; returns the number of selected items in all tiles before `tile`.
 * Every spin is bounded (SUMA_SPIN_LIMIT polls, seconds of wall time): a protocol error must surface
 * as an error code (DevState.overflow bit 3), never as a hung GPU. */
#define SUMA_SPIN_LIMIT (1u << 26)
#ifndef LB_BATCH
#define LB_BATCH 4 /* group words a lane requests before it looks at the first one */
#endif
template <bool PAIR>
__device__ Pair lookback_collect2(unsigned long long* __restrict__ status, unsigned long long* __restrict__ group,
                                  uint32_t tile, uint32_t epoch, int lane, uint32_t* __restrict__ fault) {
  const uint32_t g = tile >> 6;
  uint32_t sum = 0, sumx = 0;
  bool timed_out = false;
  /* Round 6: every word this wave needs is REQUESTED before the first one is looked at -- the status word of the lane's
   * earlier tile of the own group first, then the group accumulators LB_BATCH at a time.  Round 5 walked them one load,
   * one check at a time: two dependent memory-side round trips per tile at the 1 M map (groups, then own group), and
   * ceil(groups / 64) + 1 of them at 50 M surfels (763 groups: twelve).  A word that is not complete yet is polled as
   * before, lane by lane; the values added are the same, so is the result. */
  const uint32_t j = tile & 63u;
  const bool own = (uint32_t)lane < j;
  unsigned long long ws = 0;
  if (own) ws = __hip_atomic_load(&status[(g << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  /* complete groups before mine: every group before g holds exactly 64 tiles */
  for (uint32_t base = 0; base < g; base += 64u * LB_BATCH) {
    unsigned long long w[LB_BATCH];
#pragma unroll
    for (int q = 0; q < LB_BATCH; ++q) {
      const uint32_t gi = base + (uint32_t)q * 64u + (uint32_t)lane;
      w[q] = gi < g ? __hip_atomic_load(&group[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (64ull << 56);
    }
#pragma unroll
    for (int q = 0; q < LB_BATCH; ++q) {
      const uint32_t gi = base + (uint32_t)q * 64u + (uint32_t)lane;
      uint32_t spins = 0;
      while ((uint32_t)(w[q] >> 56) != 64u && ++spins < SUMA_SPIN_LIMIT)
        w[q] = __hip_atomic_load(&group[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      timed_out |= (spins >= SUMA_SPIN_LIMIT);
      sum += (uint32_t)(w[q] & 0xffffffffull); /* a lane beyond the last group adds the zero it was given */
      if (PAIR) sumx += (uint32_t)(w[q] >> 32) & 0xffffffu;
    }
  }
  /* earlier tiles of my own group */
  if (own) {
    uint32_t spins = 0;
    while (((uint32_t)(ws >> 34) != (epoch & 0x3fffffffu) || ((ws >> 32) & 3ull) == 0) && ++spins < SUMA_SPIN_LIMIT)
      ws = __hip_atomic_load(&status[(g << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    timed_out |= (spins >= SUMA_SPIN_LIMIT);
    sum += (uint32_t)(ws & 0xffffull);
    if (PAIR) sumx += (uint32_t)(ws >> 16) & 0xffffu;
  }
  if (timed_out) {
    atomicOr(fault, 8u);
    atomicOr(fault + (offsetof(DevState, fault_site) - offsetof(DevState, overflow)) / 4, 0x2u);
  }
  Pair r;
  r.a = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(sum), 63); /* wave total, DPP (dev_math.h) */
  r.x = PAIR ? (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(sumx), 63) : 0u;
  return r;
}
__device__ __forceinline__ uint32_t lookback_collect(unsigned long long* __restrict__ status,
                                                     unsigned long long* __restrict__ group, uint32_t tile, uint32_t epoch,
                                                     int lane, uint32_t* __restrict__ fault) {
  return lookback_collect2<false>(status, group, tile, epoch, lane, fault).a;
}
__device__ uint32_t lookback_prefix(unsigned long long* __restrict__ status, unsigned long long* __restrict__ group,
                                    uint32_t tile, uint32_t agg, uint32_t epoch, int lane, uint32_t* __restrict__ fault) {
  if (lane == 0) lookback_publish(status, group, tile, agg, epoch);
  return lookback_collect(status, group, tile, epoch, lane, fault);
}

/* block-level stable ranking of a flag: returns the rank of this thread among the block's
 * selected threads and the block total; all 256 threads call it */
struct BlockRank {
  uint32_t rank, total;
};
#define TILE_WAVES (SUMA_TILE / 64)
__device__ __forceinline__ BlockRank block_rank(bool flag, uint32_t* s_wave /* [TILE_WAVES] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long ball = __ballot(flag);
  const uint32_t below = __popcll(ball & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave[wave] = __popcll(ball);
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < (int)TILE_WAVES; ++w) {
    uint32_t cnt = s_wave[w];
    if (w < wave) off += cnt;
    tot += cnt;
  }
  BlockRank r;
  r.rank = off + below;
  r.total = tot;
  return r;
}

/* two flags ranked behind one barrier */
__device__ __forceinline__ void block_rank2(bool fa, bool fb, uint32_t* s_wa, uint32_t* s_wb, BlockRank* ra, BlockRank* rb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long ba = __ballot(fa), bb = __ballot(fb);
  const unsigned long long lo = (1ull << lane) - 1ull;
  if (lane == 0) {
    s_wa[wave] = __popcll(ba);
    s_wb[wave] = __popcll(bb);
  }
  __syncthreads();
  uint32_t offa = 0, tota = 0, offb = 0, totb = 0;
#pragma unroll
  for (int w = 0; w < (int)TILE_WAVES; ++w) {
    const uint32_t ca = s_wa[w], cb = s_wb[w];
    if (w < wave) {
      offa += ca;
      offb += cb;
    }
    tota += ca;
    totb += cb;
  }
  ra->rank = offa + __popcll(ba & lo);
  ra->total = tota;
  rb->rank = offb + __popcll(bb & lo);
  rb->total = totb;
}

/* Ticket bookkeeping shared by the compaction kernels.  Every block draws tickets until it gets
 * one >= ntiles; those failing tickets are ntiles .. ntiles + gridDim.x - 1, and a block draws
 * its failing ticket only after it has finished all of its tiles, so the block that draws the
 * largest one knows that every other block is done: it re-arms the ticket and clears the group
 * accumulators the NEXT launch will use (launches alternate between two halves, so words that may
 * still receive a straggling atomic of this launch are never written here). */
__device__ __forceinline__ bool is_finaliser(uint32_t failing_ticket, uint32_t ntiles) {
  return failing_ticket == ntiles + gridDim.x - 1;
}
__device__ __forceinline__ void finalise_tickets(DevState* ds, unsigned long long* group_next, uint32_t group_words) {
  for (uint32_t gi = threadIdx.x; gi < group_words; gi += blockDim.x) group_next[gi] = 0;
  if (threadIdx.x == 0) ds->ticket = 0;
}

/* extractSurfels, SurfelMap.cpp:725-739: the tile's block in the cache arena becomes its SubmapCache entry; n = surfels
 * selected (capacity SurfelMap.cpp:279) */
__device__ __forceinline__ void commit_extraction(DevState* ds, CacheSlot* slots, uint32_t slot, uint32_t base,
                                                  uint32_t arena_cap, uint32_t n) {
  if (n > SUMA_EXTRACT_CAPACITY) {
    n = SUMA_EXTRACT_CAPACITY;
    atomicOr(&ds->overflow, 4u);
  }
  if ((uint64_t)base + n > arena_cap) {
    n = arena_cap - base;
    atomicOr(&ds->overflow, 2u);
  }
  slots[slot].offset = base;
  slots[slot].count = n;
  ds->cache_used = base + n;
  ds->n_extracted = n;
}

/* ---------------------------------------------------------------------------------------------
 * K8 + clear of the integration mask
 * ------------------------------------------------------------------------------------------- */
/* per-update counters + SurfelMap.cpp:494-495: poses_[timestamp_] = pose */
__device__ __forceinline__ void write_pose_entry(float* poses, float* poses_inv, uint32_t pose_idx, const m4& pose) {
  float inv[16];
  rigid_inverse_dev(pose.m, inv);
  for (int i = 0; i < 16; ++i) {
    poses[16 * (size_t)pose_idx + i] = pose.m[i];
    poses_inv[16 * (size_t)pose_idx + i] = inv[i];
  }
}
__global__ void __launch_bounds__(256)
    k8_radius(const float4* __restrict__ V, const float4* __restrict__ N, K8Out o, uint32_t P, DevState* ds,
              float* poses, float* poses_inv, uint32_t pose_idx, m4 pose, const float4* __restrict__ Sem) {
  uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= P) return;
  if (pix == 0) {
    ds->n_updated = 0;
    ds->n_data = 0;
    ds->n_kept_updated = 0;
    ds->n_kept_data = 0;
    write_pose_entry(poses, poses_inv, pose_idx, pose);
  }
  k8_pixel(o, pix, V[pix], N[pix], Sem[pix]);
}
/* the pose-table part alone, for updates whose per-pixel K8 work rode on the statistics pass */
__global__ void k8_pose_only(float* poses, float* poses_inv, uint32_t pose_idx, m4 pose) {
  if (threadIdx.x == 0 && blockIdx.x == 0) write_pose_entry(poses, poses_inv, pose_idx, pose);
}

/* ---------------------------------------------------------------------------------------------
 * K7 index map: nearest visible surfel per data pixel
 * ------------------------------------------------------------------------------------------- */
struct UpdArgs {
  const suma_surfel* in;
  suma_surfel* out;
  DevState* ds;
  const float* poses;
  const float* poses_inv;
  unsigned long long* zbuf; /* data sized: K7 keys */
  unsigned long long* status;
  unsigned long long *group, *group_next; /* per 64 tiles: {tiles arrived, sum}; this launch / next launch */
  uint32_t group_words;
  uint32_t epoch;
  const float4 *V, *N, *Sem;
  const float4* radius_conf;
  const float4* pixrec; /* packed K9 gather record, 4 x float4 per pixel (3 used) */
  uint8_t* integrated;
  uint32_t* index_map;
  proj_t q;
  m4 pose, inv_pose;
  int32_t timestamp;
  uint32_t max_surfels;
  /* K9 parameters (SurfelMap.cpp:399-438) */
  float confidence_threshold, map_max_distance, update_angle_thresh;
  float p_stable, p_unstable, log_prior, log_unstable, sigma_angle, sigma_distance, max_weight;
  int32_t use_stability, unstable_age, confidence_mode, active_timestamps, weighting_scheme, averaging_scheme,
      update_always;
  /* K11 */
  float cx, cy, extent;
  /* K12 selection riding on K9 / K10: the tile that is extracted right after this update (ex_flags == NULL: none) */
  uint8_t* ex_flags;
  float ex_cx, ex_cy, ex_extent;
  /* K12 itself riding on K9 / K10 (k9_update<true>, k10_generate<true>): the flagged records are ALSO written, in
   * order, to the cache arena at DevState.cache_used; K10's finaliser commits the tile (slot table, bump pointer) */
  suma_surfel* x_arena;
  CacheSlot* x_slots;
  uint32_t x_cap, x_slot;
  /* K9 also records poses_[timestamp_] when K8 did not run as a kernel of its own */
  float* poses_w;
  float* poses_inv_w;
  uint32_t pose_idx;
  int write_pose;
};

__device__ __forceinline__ void load_pose(const float* __restrict__ table, int32_t idx, float* M) {
  const float4* src = reinterpret_cast<const float4*>(table + 16 * (size_t)idx);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 col = src[k];
    M[4 * k] = col.x;
    M[4 * k + 1] = col.y;
    M[4 * k + 2] = col.z;
    M[4 * k + 3] = col.w;
  }
}

__global__ void __launch_bounds__(256) k7_indexmap(UpdArgs a) {
  const uint32_t S = a.ds->n_surfels;
  const float4* __restrict__ sf = reinterpret_cast<const float4*>(a.in);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
    float4 s0 = sf[4 * (size_t)i], s1 = sf[4 * (size_t)i + 1];
    float count = sf[4 * (size_t)i + 2].w;
    float Ps[16], M[16];
    load_pose(a.poses, (int32_t)count, Ps);
    m4_mul(a.inv_pose.m, Ps, M);
    v3 p = m4_point(M, xyz(s0)), n = m4_dir(M, xyz(s1));
    float lp = len3(p);
    if (!(dot3(n, divs3(neg3(p), lp)) > 0.01f)) continue;
    v3 pr = project01(a.q, p);
    float fx = sdm_floor(pr.x * a.q.width), fy = sdm_floor(pr.y * a.q.height);
    if (!(fx >= 0.0f && fx < a.q.width && fy >= 0.0f && fy < a.q.height)) continue;
    float zn = 2.0f * pr.z - 1.0f;
    if (!(zn >= -1.0f && zn <= 1.0f)) continue;
    unsigned long long key = ((unsigned long long)depth24(0.5f * zn + 0.5f) << 32) | i;
    zbuf_min(&a.zbuf[(size_t)(int32_t)fy * a.q.W + (size_t)(int32_t)fx], key);
  }
}

/* update_surfels.vert:113-124 slerp().  Documented deviation: where the GLSL would
 * produce NaN (sin(omega) not > 0: identical normals) v0 is returned. */
__device__ __forceinline__ v3 slerp3(v3 v0, v3 v1, float weight) {
  float omega = sdm_acos(dot3(normalize3(v0), normalize3(v1)));
  float so = sdm_sin(omega);
  if (!(so > 0.0f)) return v0;
  float eta = 1.0f / so;
  float w0 = eta * sdm_sin(weight * omega);
  float w1 = eta * sdm_sin((1.0f - weight) * omega);
  return add3(scale3(w0, v0), scale3(w1, v1));
}

struct Surfel4 {
  float4 a, b, c, d; /* pos+radius | normal+confidence | timestamp,color,weight,count | semantic */
};

/* K11 predicate, copy_surfels.vert:38-56.  *in_tile: the K12 predicate (extract_surfels.vert:46-64) for the tile this
 * update flags -- the same world position (pose entry x record), evaluated here so that K12 need not recompute it for
 * the whole map */
__device__ __forceinline__ bool in_active_area(const UpdArgs& a, const Surfel4& s, bool* in_tile) {
  float Ps[16];
  load_pose(a.poses, (int32_t)s.c.w, Ps);
  v3 pos = m4_point(Ps, xyz(s.a));
  *in_tile = !(sdm_abs(pos.x - a.ex_cx) > a.ex_extent || sdm_abs(pos.y - a.ex_cy) > a.ex_extent);
  if ((int32_t)__float_as_uint(s.c.x) < 0) return false;
  if (sdm_abs(pos.x - a.cx) > a.extent || sdm_abs(pos.y - a.cy) > a.extent) return false;
  return true;
}

__device__ __forceinline__ void store_surfel(suma_surfel* out, uint32_t idx, const Surfel4& s) {
  float4* o = reinterpret_cast<float4*>(out) + 4 * (size_t)idx;
  store_stream(o, s.a);
  store_stream(o + 1, s.b);
  store_stream(o + 2, s.c);
  store_stream(o + 3, s.d);
}

/* K9 for one surfel, in three stages so that a lane can keep TWO surfels in flight (the stages are
 * separated by dependent loads: surfel -> pose table entry -> measurement record):
 *   k9_prepare  surfel + pose entry  -> world / sensor frame quantities, projection, texel address
 *   k9_gather   the measurement record (one 64-byte line, built by K8) + the K7 key of that pixel
 *   k9_finish   update_surfels.vert proper; returns keep, `o` receives the updated record,
 *               *mark_pix >= 0 if the surviving surfel marks that measurement pixel as integrated */
struct K9Pre {
  v3 old_position, old_normal;
  float imz;
  int32_t tx, ty;
  uint32_t rpix;
  bool in_tex, visible, inside;
};
struct K9Rec {
  float4 dv, dn, dx;
  unsigned long long k7key;
};

__device__ __forceinline__ K9Pre k9_prepare(const UpdArgs& a, const Surfel4& in) {
  K9Pre p;
  const int32_t creation_timestamp = (int32_t)in.c.w;
  float Ps[16];
  load_pose(a.poses, creation_timestamp, Ps);
  p.old_position = m4_point(Ps, xyz(in.a));
  p.old_normal = m4_dir(Ps, xyz(in.b));
  const v3 vertex = m4_point(a.inv_pose.m, p.old_position);
  const v3 normal = normalize3(m4_dir(a.inv_pose.m, p.old_normal));
  p.visible = dot3(normal, divs3(neg3(vertex), len3(vertex))) > 0.0f;
  const v3 pr = project01(a.q, vertex);
  const float imx = sdm_floor(pr.x * a.q.width) + 0.5f, imy = sdm_floor(pr.y * a.q.height) + 0.5f;
  p.imz = pr.z;
  /* texel fetch at the exact centre (imx, imy); border (0) outside or for NaN */
  p.in_tex = (imx >= 0.0f && imx < a.q.width && imy >= 0.0f && imy < a.q.height);
  p.tx = p.in_tex ? (int32_t)sdm_floor(imx) : -1;
  p.ty = p.in_tex ? (int32_t)sdm_floor(imy) : -1;
  p.rpix = (uint32_t)max(p.ty, 0) * (uint32_t)a.q.W + (uint32_t)max(p.tx, 0);
  /* quirk B-6: all(lessThan(img, dim)) && !all(lessThan(img, 0)) */
  p.inside = (imx < a.q.width && imy < a.q.height && p.imz < 1.0f) && !(imx < 0.0f && imy < 0.0f && p.imz < 0.0f);
  return p;
}

/* one 64-byte line per measurement pixel (built by K8): vertex, normal, (label, prob, radius);
 * border (0) outside the image or for NaN coordinates, as the NEAREST / CLAMP_TO_BORDER fetch */
__device__ __forceinline__ K9Rec k9_gather(const UpdArgs& a, const K9Pre& p) {
  K9Rec r;
  const float4* __restrict__ rec = a.pixrec + 4 * (size_t)p.rpix;
  r.dv = rec[0];
  r.dn = rec[1];
  r.dx = rec[2];
  r.k7key = a.zbuf[p.rpix];
  return r;
}

__device__ __forceinline__ bool k9_finish(const UpdArgs& a, uint32_t i, const Surfel4& in, const K9Pre& p,
                                          const K9Rec& g, Surfel4& o, int32_t* mark_pix) {
  const int32_t timestamp = a.timestamp;
  const int32_t W = a.q.W;
  const int32_t surfel_age = timestamp - (int32_t)__float_as_uint(in.c.x);
  const int32_t creation_timestamp = (int32_t)in.c.w;
  const v3 old_position = p.old_position, old_normal = p.old_normal;
  const float old_radius = in.a.w, old_confidence = in.b.w, old_weight = in.c.z;

  bool keep = true;
  if (old_confidence < a.confidence_threshold && a.use_stability) keep = (surfel_age < a.unstable_age);
  o = in;
  o.c.y = pack_rgb(0.3f, 0.3f, 0.3f);

  float4 dv = g.dv, dn = g.dn, dx = g.dx;
  if (!p.in_tex) dv = dn = dx = f4(0.f, 0.f, 0.f, 0.f);
  const float4 ds = f4(dx.x, 0.f, 0.f, dx.y), rc = f4(dx.z, 0.f, 0.f, 0.f);
  const unsigned long long k7key = g.k7key;
  const bool valid = (dv.w > 0.5f) && (dn.w > 0.5f);
  const float imz = p.imz;
  const int32_t tx = p.tx, ty = p.ty;

  float penalty = 0.0f;
  float update_confidence = a.log_prior;
  bool mark = false;

  if (valid && p.inside && p.visible) {
    const float data_label = ds.x * 255.0f, data_prob = ds.w;
    const float model_label = in.d.x * 255.0f, model_prob = in.d.w;
    if (sdm_round(data_label) != sdm_round(model_label)) {
      if (is_dynamic_label(model_label)) penalty = 1.0f;
    }
    const v3 v = xyz(dv), n = xyz(dn);
    const v3 v_global = m4_point(a.pose.m, v);
    const v3 n_global = normalize3(m4_dir(a.pose.m, n));
    const v3 view_dir = divs3(neg3(v), len3(v));
    const float distance = sdm_abs(dot3(old_normal, sub3(v_global, old_position)));
    const float angle = len3(cross3(n_global, old_normal));
    const float new_radius = rc.x, new_confidence = rc.y;

    if ((distance < a.map_max_distance) && (angle < a.update_angle_thresh)) {
      mark = true; /* gl_Position inside the viewport: update_surfels.vert:219 */
      const float confidence = old_confidence + new_confidence;
      o.b.w = confidence;
      o.c.x = __uint_as_float((uint32_t)timestamp);
      float avg_radius = fmin_(new_radius, old_radius);
      avg_radius = fmax_(avg_radius, 0.0f); /* the update program's min_radius uniform is 0, SurfelMap.cpp:422 */
      o.a.w = avg_radius;
      keep = true;
      o.c.y = pack_rgb(0.0f, 0.7f, 0.0f);
      o.c.w = (float)creation_timestamp;

      float pst = a.p_stable;
      if (a.confidence_mode == 1 || a.confidence_mode == 3)
        pst *= sdm_exp((-angle * angle) / (a.sigma_angle * a.sigma_angle));
      if (a.confidence_mode == 2 || a.confidence_mode == 3)
        pst *= sdm_exp((-distance * distance) / (a.sigma_distance * a.sigma_distance));
      pst = fclamp(pst, a.p_unstable, 1.0f);
      update_confidence = sdm_log(pst / (1.0f - pst));

      if ((new_radius < old_radius && timestamp - creation_timestamp < a.active_timestamps) || a.update_always) {
        float w1 = 0.9f, w2 = 0.1f;
        if (a.weighting_scheme > 0) {
          w1 = old_weight;
          w2 = 1.0f;
          if (a.weighting_scheme == 2) w2 = dot3(n, view_dir);
          o.c.z = fmin_(a.max_weight, w1 + w2);
          float sum = w1 + w2;
          w1 /= sum;
          w2 /= sum;
        }
        v3 avg_position = add3(scale3(w1, old_position), scale3(w2, v_global));
        v3 avg_normal = slerp3(old_normal, n_global, w1);
        float avg_prob;
        if (sdm_round(data_label) != sdm_round(model_label))
          avg_prob = w1 * model_prob + w2 * (1.0f - data_prob);
        else
          avg_prob = w1 * model_prob + w2 * data_prob;
        o.d.w = avg_prob;
        if (a.averaging_scheme == 1) {
          avg_position = add3(old_position, scale3(w2 * distance, old_normal));
          avg_normal = slerp3(old_normal, n_global, w1);
        }
        avg_normal = normalize3(avg_normal);
        float Pi[16];
        load_pose(a.poses_inv, creation_timestamp, Pi);
        avg_position = m4_point(Pi, avg_position);
        avg_normal = m4_dir(Pi, avg_normal);
        o.a = f4(avg_position.x, avg_position.y, avg_position.z, avg_radius);
        o.b = f4(avg_normal.x, avg_normal.y, avg_normal.z, confidence);
        o.c.y = pack_rgb(1.0f, 0.0f, 1.0f);
      }
    } else {
      /* K7 winner of this measurement pixel */
      int32_t idx = (k7key == SUMA_EMPTY_KEY) ? -1 : (int32_t)(uint32_t)(k7key & 0xffffffffull);
      if (idx == (int32_t)i) {
        update_confidence = sdm_log(a.p_unstable / (1.0f - a.p_unstable));
        o.c.y = pack_rgb(0.0f, 1.0f, 1.0f);
      }
    }
  }
  update_confidence = update_confidence - penalty;
  if (a.use_stability)
    o.b.w = fmin_((old_confidence + update_confidence) - a.log_prior, 20.0f);
  else
    o.b.w = old_confidence;
  if (o.b.w < a.log_unstable && a.use_stability) keep = false;

  *mark_pix = -1;
  if (keep && mark) {
    /* the rasterised point of the surviving surfel marks the measurement as integrated; it is
     * clipped unless z_ndc = 2*z01 - 1 lies in [-1, 1] */
    float zn = 2.0f * imz - 1.0f;
    if (zn >= -1.0f && zn <= 1.0f) *mark_pix = ty * W + tx;
  }
  return keep;
}

/* K9 (+ K11 predicate): single-pass update with stable compaction, tiles of SUMA_TILE surfels handed
 * out by ticket, output offset by the two-level look-back above.
 *
 * The per-surfel work is a chain of dependent memory round trips (surfel -> pose entry -> measurement
 * record) with divergent arithmetic at the end, so the kernel is organised for latency, not bandwidth:
 *  - the three stages of a surfel are separate functions, so a lane can run K9_PER surfels with their
 *    stages interleaved (K9_PER x the loads in flight per wave).  Measured: 2 x 512 threads and
 *    1 x 1024 threads are equal at 1 M surfels; at 50 M surfels 16 waves per CU win (2.06 vs 1.73 TB/s),
 *    hence K9_PER = 1;
 *  - a lane drops its updated record into LDS at its own (uncompacted) slot as soon as it is computed;
 *    the stable ranks follow from ballots and a rank -> slot table, and the stream-out walks that table,
 *    writing a dense 16 B-per-lane stream instead of 64-byte-strided record stores;
 *  - the records are double buffered in LDS (2 x 64 KB): a tile's output offset depends on every earlier
 *    tile of the launch having published its count, so it is collected only after the NEXT tile has been
 *    computed and published -- by then the words are almost always there (waiting right after the compute
 *    exposes each block to the slowest of the ~256 tiles in flight ahead of it);
 *  - the barriers are LDS-only (no vmcnt drain): the fire-and-forget stores (integration mask, status
 *    words, the previous tile's stream-out) stay in flight across them.
 * Round 4 built the same kernel WITHOUT block barriers (every wave owns its 64 surfels from load to stream-out, counts
 * and offsets exchanged through polled LDS words; git log -S k9_update_w): bit-identical and slower, 88 us against 76 --
 * a tile's total cannot be published before its slowest wave has reported, so the drift the barriers prevent comes
 * back one for one as look-back wait of every later tile (profiles/r04_k9_wave_independent_experiment.txt). */
#ifndef K9_PER
#define K9_PER 1 /* surfels per lane (see above) */
#endif
#define K9_THREADS (SUMA_TILE / K9_PER)
#define K9_WAVES (K9_THREADS / 64)
static_assert(SUMA_TILE % K9_PER == 0 && K9_THREADS <= 1024, "K9 tile split");

__device__ __forceinline__ Surfel4 load_surfel(const float4* __restrict__ sf, uint32_t i) {
  Surfel4 r;
  r.a = sf[4 * (size_t)i];
  r.b = sf[4 * (size_t)i + 1];
  r.c = sf[4 * (size_t)i + 2];
  r.d = sf[4 * (size_t)i + 3];
  return r;
}

/* K9 needs ~110 launch-uniform values (two 4x4 matrices, the projection, 22 update parameters, 16 pointers) next to the
 * exec masks of update_surfels.vert's nested branches: more than the 102 SGPRs of a wave.  Holding them all for the
 * whole kernel made the compiler park 164 of them in VGPR lanes and fetch them back with a v_readlane at every use --
 * 414 VALU instructions (+ 140 hazard nops) in the loop body of a kernel whose compute phase is VALU-bound.  Instead
 * each phase of a trip re-reads what it needs from the kernel-argument segment (scalar loads through the constant
 * cache: the scalar memory unit, not the VALU); the empty asm makes the pointer opaque so that the loads stay inside
 * the phase instead of being hoisted out of the loop again. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UPD_NO_ARG_RELOAD)
#define UPD_ARG_RELOAD 1
#endif
#ifdef UPD_ARG_RELOAD
typedef const UpdArgs __attribute__((address_space(4))) * UpdArgsK;
__device__ __forceinline__ UpdArgsK upd_args_again(UpdArgsK p) {
  asm volatile("" : "+s"(p));
  return p;
}
#endif

#ifdef SUMA_PHASE_TIMING
__device__ unsigned long long g_k9_phase[PH_BLOCKS][9];
/* host: PH_BLOCKS x 9 words (8 phase totals in 10 ns units + the number of launches the block took part in) */
extern "C" int suma_debug_k9_phases(unsigned long long* host, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k9_phase), sizeof(g_k9_phase));
  if (e == hipSuccess && reset) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_k9_phase));
    if (e == hipSuccess) e = hipMemset(p, 0, sizeof(g_k9_phase));
  }
  return (int)e;
}
#endif

/* K9_DEFER 1: one block per CU, records double buffered, a tile's offset collected one tile later (the default).
 * K9_DEFER 0: records single buffered (68 KB), so that TWO blocks share a CU (with K9_PER = 2: 2 x 512 threads); a
 * block waits for its offset right after publishing, and the other block's work fills the wait. */
#ifndef K9_DEFER
#define K9_DEFER 1
#endif
#define K9_BUFS (K9_DEFER ? 2 : 1)
/* XF: the extraction that follows this update (K12, one tile) rides on the stream-out -- the flagged records get a
 * second stable rank (their place in the tile's cache block) from the same ballots, barrier and look-back words, and are
 * written to the cache arena next to their place in the map; K12 would copy exactly these records in exactly this
 * order one launch later (launch + ticket + look-back for 1 byte per surfel: 14 us on every second scan). */
template <bool XF>
__global__ void __launch_bounds__(K9_THREADS)
#if !K9_DEFER
    __attribute__((amdgpu_waves_per_eu(2 * K9_WAVES / 4, 2 * K9_WAVES / 4)))
#endif
    k9_update(UpdArgs a) {
  __shared__ float4 s_out[K9_BUFS][SUMA_TILE][4]; /* updated records at their uncompacted slot, double buffered */
  __shared__ uint16_t s_slot[K9_BUFS][SUMA_TILE]; /* stable rank -> slot */
  __shared__ uint8_t s_ext[K9_BUFS][SUMA_TILE];   /* slot -> "in the tile that is extracted after this update" */
  __shared__ uint16_t s_xrank[XF ? K9_BUFS : 1][XF ? SUMA_TILE : 1]; /* slot -> rank among the tile's extracted records */
  __shared__ uint32_t s_cnt_emit[K9_PER][K9_WAVES], s_cnt_keep[K9_WAVES], s_cnt_x[K9_PER][K9_WAVES];
  __shared__ uint32_t s_tile, s_prefix, s_prefix_x;
  const uint32_t xbase = XF ? a.ds->cache_used : 0u; /* stable during the update: K10's finaliser moves it */
  float4* __restrict__ xdst4 = reinterpret_cast<float4*>(a.x_arena);
  const uint32_t S = a.ds->n_surfels;
  const uint32_t ntiles = (S + SUMA_TILE - 1) / SUMA_TILE;
  const float4* __restrict__ sf = reinterpret_cast<const float4*>(a.in);
  float4* __restrict__ dst4 = reinterpret_cast<float4*>(a.out);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t keep_count = 0; /* thread 0: survivors before the area filter (S') */
  if (a.write_pose && blockIdx.x == 0 && threadIdx.x == 0) write_pose_entry(a.poses_w, a.poses_inv_w, a.pose_idx, a.pose);
  /* the tile whose records wait in LDS for their output offset */
  uint32_t prev_tile = 0xffffffffu, prev_total = 0, prev_totalx = 0, buf = 0;
#ifdef UPD_ARG_RELOAD
  const UpdArgsK ka = (UpdArgsK)__builtin_amdgcn_kernarg_segment_ptr(); /* `a` is the kernel's only argument */
#endif
  /* compacted stream-out of the tile in buffer pb: chunk c = (rank, 16-byte part) */
  auto stream_out = [&](uint32_t pb, uint32_t tile_id, uint32_t count, uint32_t countx) {
    const uint32_t prefix = s_prefix, prefix_x = XF ? s_prefix_x : 0u;
    for (uint32_t c = threadIdx.x; c < 4u * count; c += K9_THREADS) {
      const uint64_t d = 4ull * prefix + c;
      if (d < 4ull * a.max_surfels) {
        const uint32_t slot = s_slot[pb][c >> 2];
        const float4 v = s_out[pb][slot][c & 3u];
        store_stream(&dst4[d], v);
        if (XF) {
          if (s_ext[pb][slot]) {
            const uint32_t xr = prefix_x + s_xrank[pb][slot];
            if (xr < SUMA_EXTRACT_CAPACITY && (uint64_t)xbase + xr < a.x_cap)
              store_stream(&xdst4[4ull * ((uint64_t)xbase + xr) + (c & 3u)], v);
          }
        } else if (a.ex_flags != nullptr && (c & 3u) == 0) {
          a.ex_flags[d >> 2] = s_ext[pb][slot];
        }
      }
    }
    if (tile_id == ntiles - 1 && threadIdx.x == 0) {
      const uint32_t tot = prefix + count;
      a.ds->n_kept_updated = tot < a.max_surfels ? tot : a.max_surfels;
      if (XF) a.ds->n_ext_update = prefix_x + countx;
    }
  };
  PH_BEGIN;
  for (;;) {
    PH(7); /* loop overhead */
    lds_barrier(); /* s_tile / s_prefix / s_cnt_* of the previous trip have been read */
    if (threadIdx.x == 0)
      s_tile = __hip_atomic_fetch_add(&a.ds->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_barrier();
    const uint32_t tile = s_tile;
    PH(0); /* ticket: atomic round trip between two barriers */
    uint32_t total = 0, totalx = 0;
    if (tile < ntiles) {
      /* lane t handles slots t, K9_THREADS + t, ...: slot order = surfel order = stable order */
      uint32_t idx[K9_PER];
      bool live[K9_PER], emit[K9_PER];
      Surfel4 in[K9_PER];
      K9Pre pre[K9_PER];
      K9Rec rec[K9_PER];
#pragma unroll
      for (int u = 0; u < K9_PER; ++u) {
        idx[u] = tile * SUMA_TILE + (uint32_t)u * K9_THREADS + threadIdx.x;
        live[u] = idx[u] < S;
        in[u] = load_surfel(sf, live[u] ? idx[u] : 0u); /* out-of-range lanes read surfel 0 (S > 0 here), masked below */
      }
#ifdef UPD_ARG_RELOAD
      const UpdArgs aA = *upd_args_again(ka); /* this phase's arguments, fetched here */
#else
      const UpdArgs& aA = a;
#endif
#pragma unroll
      for (int u = 0; u < K9_PER; ++u) pre[u] = k9_prepare(aA, in[u]);
#ifdef SUMA_PHASE_TIMING
      if (__float_as_uint(in[0].a.x) == 0x7fc12345u) ph_acc[7] += 1; /* consumes the surfel loads before the stamp */
#endif
      PH(1); /* surfel loads + prepare issued */
#pragma unroll
      for (int u = 0; u < K9_PER; ++u) rec[u] = k9_gather(aA, pre[u]);
#ifdef UPD_ARG_RELOAD
      const UpdArgs aB = *upd_args_again(ka);
#else
      const UpdArgs& aB = a;
#endif
      uint32_t kept = 0;
      unsigned long long eb[K9_PER], xb[K9_PER];
      bool xt[K9_PER];
#pragma unroll
      for (int u = 0; u < K9_PER; ++u) {
        Surfel4 o;
        int32_t mark_pix;
        const bool keep = k9_finish(aB, idx[u], in[u], pre[u], rec[u], o, &mark_pix) && live[u];
        if (keep && mark_pix >= 0) aB.integrated[mark_pix] = 1;
        bool in_tile;
        emit[u] = in_active_area(aB, o, &in_tile) && keep;
        const uint32_t slot = (uint32_t)u * K9_THREADS + threadIdx.x;
        xt[u] = in_tile && emit[u];
        xb[u] = XF ? __ballot(xt[u]) : 0ull;
        s_ext[buf][slot] = in_tile ? 1 : 0;
        s_out[buf][slot][0] = o.a;
        s_out[buf][slot][1] = o.b;
        s_out[buf][slot][2] = o.c;
        s_out[buf][slot][3] = o.d;
        eb[u] = __ballot(emit[u]);
        kept += __popcll(__ballot(keep));
      }
      PH(2); /* pose entry + measurement gather + update arithmetic + LDS record */
      if (lane == 0) {
        s_cnt_keep[wave] = kept; /* S' statistics (parity with the reference's TF count) */
#pragma unroll
        for (int u = 0; u < K9_PER; ++u) {
          s_cnt_emit[u][wave] = __popcll(eb[u]);
          if (XF) s_cnt_x[u][wave] = __popcll(xb[u]);
        }
      }
      lds_barrier();
      uint32_t kc = 0, base = 0, basex = 0;
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int u = 0; u < K9_PER; ++u) {
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < K9_WAVES; ++w) {
          const uint32_t cnt = s_cnt_emit[u][w];
          if (w < wave) off += cnt;
          tot += cnt;
        }
        if (emit[u]) s_slot[buf][base + off + __popcll(eb[u] & below)] = (uint16_t)((uint32_t)u * K9_THREADS + threadIdx.x);
        base += tot;
        if (XF) {
          uint32_t offx = 0, totx = 0;
#pragma unroll
          for (int w = 0; w < K9_WAVES; ++w) {
            const uint32_t cnt = s_cnt_x[u][w];
            if (w < wave) offx += cnt;
            totx += cnt;
          }
          if (xt[u]) s_xrank[buf][(uint32_t)u * K9_THREADS + threadIdx.x] = (uint16_t)(basex + offx + __popcll(xb[u] & below));
          basex += totx;
        }
      }
#pragma unroll
      for (int w = 0; w < K9_WAVES; ++w) kc += s_cnt_keep[w];
      total = base;
      totalx = basex;
      if (threadIdx.x == 0) {
        keep_count += kc;
        lookback_publish(a.status, a.group, tile, total, a.epoch, totalx);
      }
      PH(3); /* ranks (one barrier) + publish */
    }
    /* the PREVIOUS tile's offset: every tile before it was drawn before it and is published without
     * any wait in between, and this block has published everything it holds -- no circular wait; by
     * now (one tile's compute later) the words are almost always there */
    if (prev_tile != 0xffffffffu) {
      if (threadIdx.x < 64) {
        const Pair pre = lookback_collect2<XF>(a.status, a.group, prev_tile, a.epoch, threadIdx.x, &a.ds->overflow);
        if (threadIdx.x == 0) {
          s_prefix = pre.a;
          if (XF) s_prefix_x = pre.x;
        }
      }
      lds_barrier();
      PH(4); /* look-back collect of the previous tile + barrier */
      stream_out(buf ^ 1u, prev_tile, prev_total, prev_totalx);
      PH(5); /* stream-out of the previous tile issued */
    }
    if (tile >= ntiles) break;
    prev_tile = tile;
    prev_total = total;
    prev_totalx = totalx;
#if K9_DEFER
    buf ^= 1u;
#else
    /* single buffer: this tile's offset and stream-out right away */
    if (threadIdx.x < 64) {
      const Pair pre = lookback_collect2<XF>(a.status, a.group, prev_tile, a.epoch, threadIdx.x, &a.ds->overflow);
      if (threadIdx.x == 0) {
        s_prefix = pre.a;
        if (XF) s_prefix_x = pre.x;
      }
    }
    lds_barrier();
    stream_out(0u, prev_tile, prev_total, prev_totalx);
    prev_tile = 0xffffffffu;
#endif
  }
  PH_END(g_k9_phase);
  if (threadIdx.x == 0 && keep_count)
    __hip_atomic_fetch_add(&a.ds->n_updated, keep_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (is_finaliser(s_tile, ntiles)) {
    finalise_tickets(a.ds, a.group_next, a.group_words);
    if (threadIdx.x == 0 && ntiles == 0) {
      a.ds->n_kept_updated = 0;
      if (XF) a.ds->n_ext_update = 0;
    }
  }
}

/* K10 (+ K11 predicate): new surfels for valid, not yet integrated, front-facing measurement
 * pixels, in the order of vbo_img_coords_ (x-major, SurfelMap.cpp:88-92).  Appends behind the
 * survivors of K9.  Also exports the K7 winners as a uint32 index map and leaves the z-buffer
 * cleared for the next user. */
template <bool XF>
__global__ void __launch_bounds__(SUMA_TILE) k10_generate(UpdArgs a) {
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_wave_a[TILE_WAVES], s_wave_b[TILE_WAVES], s_wave_x[TILE_WAVES];
  __shared__ uint32_t s_prefix, s_prefix_x;
  __shared__ uint8_t s_flag[SUMA_TILE];
  __shared__ uint32_t s_rank[SUMA_TILE];
  const int32_t W = a.q.W, H = a.q.H;
  const uint32_t P = (uint32_t)W * (uint32_t)H;
  const uint32_t ntiles = (P + SUMA_TILE - 1) / SUMA_TILE;
  const uint32_t base = a.ds->n_kept_updated;
  const uint32_t xbase = XF ? a.ds->cache_used : 0u;      /* moved by the finaliser only */
  const uint32_t xfirst = XF ? a.ds->n_ext_update : 0u;   /* K9's extracted records come first (map order) */
  const float color = pack_rgb(0.0f, 0.0f, 1.0f);
  uint32_t new_count = 0;
#ifdef UPD_ARG_RELOAD
  const UpdArgsK ka = (UpdArgsK)__builtin_amdgcn_kernarg_segment_ptr();
#endif
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0)
      s_tile = __hip_atomic_fetch_add(&a.ds->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= ntiles) break;
#ifdef UPD_ARG_RELOAD
    const UpdArgs aT = *upd_args_again(ka); /* the tile's arguments come from the argument segment again (see k9_update) */
#else
    const UpdArgs& aT = a;
#endif
    /* A tile is SUMA_TILE consecutive items of the x-major emission order, i.e. (when H divides
     * SUMA_TILE) a patch of SUMA_TILE / H whole image columns.  Lanes walk the patch ROW-major so
     * that the map reads are 16-texel segments instead of one texel per 32 KiB row stride; the
     * x-major rank every lane needs for the stable compaction is obtained by exchanging the flags
     * through LDS. */
    const bool patch = (SUMA_TILE % (uint32_t)H) == 0;
    const uint32_t cols = patch ? SUMA_TILE / (uint32_t)H : 1u;
    uint32_t q; /* x-major index of this lane's pixel inside the tile */
    if (patch) {
      const uint32_t yy = threadIdx.x / cols, xx = threadIdx.x - yy * cols;
      q = xx * (uint32_t)H + yy;
    } else {
      q = threadIdx.x;
    }
    const uint32_t t = tile * SUMA_TILE + q; /* x-major item index */
    bool gen = false, emit = false, in_tile = false;
    Surfel4 s;
    if (t < P) {
      const int32_t x = (int32_t)(t / (uint32_t)H), y = (int32_t)(t % (uint32_t)H);
      const size_t pix = (size_t)y * W + x;
      const float4 v = aT.V[pix], n = aT.N[pix], rc = aT.radius_conf[pix];
      /* export + clear of the K7 z-buffer */
      const unsigned long long key = aT.zbuf[pix];
      aT.index_map[pix] = (key == SUMA_EMPTY_KEY) ? 0u : (uint32_t)(key & 0xffffffffull) + 1u;
      aT.zbuf[pix] = SUMA_EMPTY_KEY;
      bool invalid = (v.w < 1.0f) || (n.w < 1.0f);
      invalid = invalid || (rc.w < 0.5f);
      const bool integrated = aT.integrated[pix] != 0;
      const v3 vv = xyz(v), nn = xyz(n);
      const v3 view_dir = divs3(neg3(vv), len3(vv));
      gen = (!invalid && !integrated && (dot3(nn, view_dir) > 0.01f));
      if (gen) {
        const v3 ng = normalize3(nn);
        const float4 sem = aT.Sem[pix];
        float conf = aT.log_prior;
        if (is_dynamic_label(sem.x * 255.0f)) conf = aT.log_prior - 0.5f;
        s.a = f4(v.x, v.y, v.z, rc.x);
        s.b = f4(ng.x, ng.y, ng.z, conf);
        s.c = f4(__uint_as_float((uint32_t)aT.timestamp), color, 1.0f, (float)aT.timestamp);
        s.d = sem;
        emit = in_active_area(aT, s, &in_tile);
      }
    }
    {
      const unsigned long long gb = __ballot(gen);
      if ((threadIdx.x & 63) == 0) s_wave_b[threadIdx.x >> 6] = __popcll(gb);
    }
    /* rank in x-major order: flags -> LDS[q], ranked by the lane whose id is the x-major index */
    s_flag[q] = (emit ? 1 : 0) | ((XF && emit && in_tile) ? 2 : 0);
    __syncthreads();
    BlockRank br, brx;
    if (XF) {
      block_rank2((s_flag[threadIdx.x] & 1) != 0, (s_flag[threadIdx.x] & 2) != 0, s_wave_a, s_wave_x, &br, &brx);
      s_rank[threadIdx.x] = br.rank | (brx.rank << 16);
    } else {
      br = block_rank(s_flag[threadIdx.x] != 0, s_wave_a);
      brx.rank = brx.total = 0;
      s_rank[threadIdx.x] = br.rank;
    }
    if (threadIdx.x == 0)
      for (int w = 0; w < (int)TILE_WAVES; ++w) new_count += s_wave_b[w];
    if (threadIdx.x < 64) {
      if (threadIdx.x == 0) lookback_publish(aT.status, aT.group, tile, br.total, aT.epoch, brx.total);
      const Pair pre = lookback_collect2<XF>(aT.status, aT.group, tile, aT.epoch, threadIdx.x, &aT.ds->overflow);
      if (threadIdx.x == 0) {
        s_prefix = pre.a;
        if (XF) s_prefix_x = pre.x;
      }
    }
    __syncthreads();
    if (emit) {
      const uint32_t rk = s_rank[q];
      uint64_t dst = (uint64_t)base + s_prefix + (XF ? (rk & 0xffffu) : rk);
      if (dst < aT.max_surfels) {
        store_surfel(aT.out, (uint32_t)dst, s);
        if (XF) {
          if (in_tile) {
            const uint64_t xr = (uint64_t)xfirst + s_prefix_x + (rk >> 16);
            if (xr < SUMA_EXTRACT_CAPACITY && (uint64_t)xbase + xr < aT.x_cap) store_surfel(aT.x_arena, (uint32_t)(xbase + xr), s);
          }
        } else if (aT.ex_flags != nullptr) {
          aT.ex_flags[dst] = in_tile ? 1 : 0;
        }
      }
    }
    if (tile == ntiles - 1 && threadIdx.x == 0) {
      uint64_t kept = (uint64_t)s_prefix + br.total;
      uint64_t total = (uint64_t)base + kept;
      aT.ds->n_kept_data = (uint32_t)kept;
      if (total > aT.max_surfels) {
        total = aT.max_surfels;
        atomicOr(&aT.ds->overflow, 1u);
      }
      aT.ds->n_surfels = (uint32_t)total; /* the compaction target becomes the active map */
      if (XF) aT.ds->n_extracted = xfirst + s_prefix_x + brx.total;
    }
  }
  if (threadIdx.x == 0 && new_count)
    __hip_atomic_fetch_add(&a.ds->n_data, new_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (is_finaliser(s_tile, ntiles)) {
    finalise_tickets(a.ds, a.group_next, a.group_words);
    if (XF && threadIdx.x == 0) commit_extraction(a.ds, a.x_slots, a.x_slot, xbase, a.x_cap, a.ds->n_extracted);
  }
}

static void set_m4(m4& d, const float* s) {
  for (int i = 0; i < 16; ++i) d.m[i] = s[i];
}

static uint32_t compact_grid(suma_ctx* c, uint64_t items) {
  uint64_t blocks = (items + SUMA_TILE - 1) / SUMA_TILE;
  if (blocks > SUMA_COMPACT_BLOCKS) blocks = SUMA_COMPACT_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (uint32_t)blocks;
}
static uint32_t stream_grid(suma_ctx* c, uint64_t items) {
  uint64_t blocks = (items + 255) / 256;
  if (blocks > SUMA_STREAM_BLOCKS) blocks = SUMA_STREAM_BLOCKS;
  if (blocks < 256) blocks = 256;
  return (uint32_t)blocks;
}

K8Out launch_k8_out(suma_ctx* c) {
  K8Out o;
  o.radius_conf = c->radius_conf;
  o.integrated = c->integrated;
  o.pixrec = c->pixrec;
  o.pixel_size = c->mc.pixel_size;
  o.angle_thresh = c->mc.radconf_angle_thresh;
  o.min_radius = c->p.min_radius;
  o.max_radius = c->p.max_radius;
  return o;
}

/* K7..K11 of SurfelMap::update for the current map (c->surfels[c->cur]); the result lands in the
 * other buffer, which the caller makes current. */
hipError_t launch_map_update(suma_ctx* c, const float* pose, const float* inv_pose, const suma_frame* f, float cx,
                             float cy, float extent, int k7_done, const float* ex, int fused_slot) {
  const uint32_t P = (uint32_t)c->P;
  const double S = (double)c->known_surfels;
  UpdArgs a;
  memset(&a, 0, sizeof(a));
  a.in = c->surfels[c->cur];
  a.out = c->surfels[c->cur ^ 1];
  a.ds = c->ds;
  a.poses = c->poses;
  a.poses_inv = c->poses_inv;
  a.poses_w = c->poses;
  a.poses_inv_w = c->poses_inv;
  a.zbuf = c->zbuf_data;
  a.status = c->tile_status;
  a.group_words = c->group_words;
  a.V = f->map[SUMA_MAP_VERTEX];
  a.N = f->map[SUMA_MAP_NORMAL];
  a.Sem = f->map[SUMA_MAP_SEMANTIC];
  a.radius_conf = c->radius_conf;
  a.pixrec = c->pixrec;
  a.integrated = c->integrated;
  a.index_map = c->index_map;
  a.q = c->pd;
  set_m4(a.pose, pose);
  set_m4(a.inv_pose, inv_pose);
  a.timestamp = (int32_t)c->timestamp;
  a.max_surfels = c->p.max_surfels;
  a.confidence_threshold = c->p.confidence_threshold;
  a.map_max_distance = c->p.map_max_distance;
  a.update_angle_thresh = c->mc.update_angle_thresh;
  a.p_stable = c->p.p_stable;
  a.p_unstable = c->mc.p_unstable;
  a.log_prior = c->mc.log_prior;
  a.log_unstable = c->mc.log_unstable;
  a.sigma_angle = c->p.sigma_angle;
  a.sigma_distance = c->p.sigma_distance;
  a.max_weight = c->p.max_weight;
  a.use_stability = c->p.use_stability;
  a.unstable_age = c->p.unstable_age;
  a.confidence_mode = c->p.confidence_mode;
  a.active_timestamps = c->p.active_timestamps;
  a.weighting_scheme = c->p.weighting_scheme;
  a.averaging_scheme = c->p.averaging_scheme;
  a.update_always = c->p.update_always;
  a.cx = cx;
  a.cy = cy;
  a.extent = extent;
  a.ex_flags = (ex && fused_slot < 0) ? c->extract_flags : nullptr;
  const bool xf = ex && fused_slot >= 0;
  a.x_arena = xf ? c->cache_arena : nullptr;
  a.x_slots = c->cache_slots;
  a.x_cap = c->cache_cap;
  a.x_slot = xf ? (uint32_t)fused_slot : 0u;
  a.ex_cx = ex ? ex[0] : 0.0f;
  a.ex_cy = ex ? ex[1] : 0.0f;
  a.ex_extent = ex ? ex[2] : 0.0f;
  hipStream_t st = c->ls;
  const uint32_t gridS = stream_grid(c, (uint64_t)c->known_surfels + 2 * c->P);
  a.pose_idx = c->timestamp;
  a.write_pose = 0;
  if (c->k8_fused_frame == f && c->k8_fused_stamp == c->timestamp && c->k8_fused_params == c->params_version) {
    /* the per-pixel K8 products and the counter resets were produced by the statistics pass that streamed
     * this frame (suma_api.hip, update_pose); what is left is the pose-table entry, which K9's first
     * thread writes (nothing in K9 reads the entry of the current stamp) */
    a.write_pose = 1;
  } else {
    ProfScope ps(c, "k8_radius", 48.0 * P);
    k8_radius<<<(P + 255) / 256, 256, 0, st>>>(a.V, a.N, launch_k8_out(c), P, c->ds, c->poses, c->poses_inv, c->timestamp,
                                                a.pose, a.Sem);
  }
  c->k8_fused_frame = nullptr;
  if (!k7_done) { /* otherwise the splat was fused into the post-ICP render pass (same pose, same map) */
    ProfScope ps(c, "k7_indexmap", 64.0 * S + 8.0 * P);
    k7_indexmap<<<gridS, 256, 0, st>>>(a);
  }
  {
    ProfScope ps(c, "k9_update_surfels", 128.0 * S + 16.0 * P);
    a.epoch = ++c->epoch;
    a.group = c->tile_group + (size_t)(a.epoch & 1u) * c->group_words;
    a.group_next = c->tile_group + (size_t)((a.epoch + 1u) & 1u) * c->group_words;
    const uint32_t grid = compact_grid(c, (uint64_t)c->known_surfels + 2 * c->P);
    if (xf)
      k9_update<true><<<grid, K9_THREADS, 0, st>>>(a);
    else
      k9_update<false><<<grid, K9_THREADS, 0, st>>>(a);
  }
  {
    ProfScope ps(c, "k10_generate_surfels", (80.0 + 12.0) * P + 64.0 * P * 0.5);
    a.epoch = ++c->epoch;
    a.group = c->tile_group + (size_t)(a.epoch & 1u) * c->group_words;
    a.group_next = c->tile_group + (size_t)((a.epoch + 1u) & 1u) * c->group_words;
    if (xf)
      k10_generate<true><<<compact_grid(c, P), SUMA_TILE, 0, st>>>(a);
    else
      k10_generate<false><<<compact_grid(c, P), SUMA_TILE, 0, st>>>(a);
  }
  return hipGetLastError();
}

/* ---------------------------------------------------------------------------------------------
 * pose table (SurfelMap.cpp:494-495, 485-490): poses and their rigid inverses
 * ------------------------------------------------------------------------------------------- */
__global__ void k_set_poses(float* poses, float* poses_inv, const float* src, uint32_t first, uint32_t n) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float m[16], inv[16];
  for (int i = 0; i < 16; ++i) m[i] = src[16 * (size_t)k + i];
  rigid_inverse_dev(m, inv);
  for (int i = 0; i < 16; ++i) {
    poses[16 * (size_t)(first + k) + i] = m[i];
    poses_inv[16 * (size_t)(first + k) + i] = inv[i];
  }
}
__global__ void k_identity_poses(float* poses, float* poses_inv, uint32_t n) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  for (int i = 0; i < 16; ++i) {
    float v = (i % 5 == 0) ? 1.0f : 0.0f;
    poses[16 * (size_t)k + i] = v;
    poses_inv[16 * (size_t)k + i] = v;
  }
}

__global__ void k_clear_keys(unsigned long long* z, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) z[i] = SUMA_EMPTY_KEY;
}
hipError_t launch_clear_index_zbuf(suma_ctx* c) {
  uint32_t P = (uint32_t)c->P;
  k_clear_keys<<<(P + 255) / 256, 256, 0, c->ls>>>(c->zbuf_data, P);
  return hipGetLastError();
}

hipError_t launch_set_poses(suma_ctx* c, const float* d_src, uint32_t first, uint32_t n) {
  if (n == 0) return hipSuccess;
  k_set_poses<<<(n + 255) / 256, 256, 0, c->ls>>>(c->poses, c->poses_inv, d_src, first, n);
  return hipGetLastError();
}
hipError_t launch_fill_identity_poses(suma_ctx* c) {
  uint32_t n = c->p.max_poses;
  k_identity_poses<<<(n + 255) / 256, 256, 0, c->ls>>>(c->poses, c->poses_inv, n);
  return hipGetLastError();
}

/* ---------------------------------------------------------------------------------------------
 * K12 + submap cache (device arena)
 * ------------------------------------------------------------------------------------------- */
struct ExtractArgs {
  const suma_surfel* in;
  suma_surfel* arena;
  DevState* ds;
  CacheSlot* slots;
  const float* poses;
  unsigned long long* status;
  unsigned long long *group, *group_next;
  uint32_t group_words;
  uint32_t epoch, slot, arena_cap;
  float cx, cy, extent;
  const uint8_t* flags; /* selection bytes of the update that has just run (K9 / K10 / k_append_cached), or NULL */
};

/* extract_surfels.vert:46-64: stable compaction of the surfels of one submap tile into the
 * cache arena (capacity SUMA_EXTRACT_CAPACITY per tile, SurfelMap.cpp:279).  The selection is sparse
 * (one 20 m tile out of a 180 m window), so a block takes K12_ITEMS x 1024 surfels per ticket with all
 * of its loads in flight at once, decides from position + creation stamp only, and re-reads the few
 * selected records when their output offset is known. */
#define K12_ITEMS 4u
/* FLAGS: the selection was made by the update that has just run (one byte per surfel at its final index, written by
 * K9 / K10 and, for tiles re-appended from the cache, by k_append_cached): one byte per surfel is read instead of
 * position + creation stamp + pose entry */
template <bool FLAGS>
__global__ void __launch_bounds__(SUMA_TILE) k12_extract(ExtractArgs a) {
  __shared__ uint32_t s_tile, s_prefix;
  __shared__ uint32_t s_cnt[K12_ITEMS][TILE_WAVES];
  const uint32_t S = a.ds->n_surfels;
  const uint32_t span = SUMA_TILE * K12_ITEMS;
  const uint32_t ntiles = (S + span - 1) / span;
  const float4* __restrict__ sf = reinterpret_cast<const float4*>(a.in);
  const uint32_t base = a.ds->cache_used; /* stable during the kernel: only the finaliser moves it */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0)
      s_tile = __hip_atomic_fetch_add(&a.ds->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= ntiles) break;
    float4 pa[K12_ITEMS];
    float cnt[K12_ITEMS];
    uint32_t fl[K12_ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < K12_ITEMS; ++k) { /* all loads first */
      const uint32_t i = tile * span + k * SUMA_TILE + threadIdx.x;
      pa[k] = f4(0.f, 0.f, 0.f, 0.f);
      cnt[k] = 0.f;
      fl[k] = 0;
      if (i < S) {
        if (FLAGS) {
          fl[k] = a.flags[i];
        } else {
          pa[k] = sf[4 * (size_t)i];
          cnt[k] = sf[4 * (size_t)i + 2].w;
        }
      }
    }
    bool sel[K12_ITEMS];
    uint32_t below[K12_ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < K12_ITEMS; ++k) {
      const uint32_t i = tile * span + k * SUMA_TILE + threadIdx.x;
      sel[k] = false;
      if (FLAGS) {
        sel[k] = fl[k] != 0; /* 0 beyond S */
      } else if (i < S) {
        float Ps[16];
        load_pose(a.poses, (int32_t)cnt[k], Ps);
        v3 pos = m4_point(Ps, xyz(pa[k]));
        sel[k] = !(sdm_abs(pos.x - a.cx) > a.extent || sdm_abs(pos.y - a.cy) > a.extent);
      }
      const unsigned long long b = __ballot(sel[k]);
      below[k] = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) s_cnt[k][wave] = __popcll(b);
    }
    __syncthreads();
    uint32_t total = 0, rank[K12_ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < K12_ITEMS; ++k) { /* item order = (k, wave, lane) = input order */
      uint32_t off = total;
#pragma unroll
      for (int w = 0; w < (int)TILE_WAVES; ++w) {
        const uint32_t c = s_cnt[k][w];
        if (w < wave) off += c;
        total += c;
      }
      rank[k] = off + below[k];
    }
    if (threadIdx.x < 64) {
      uint32_t pre = lookback_prefix(a.status, a.group, tile, total, a.epoch, threadIdx.x, &a.ds->overflow);
      if (threadIdx.x == 0) s_prefix = pre;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < K12_ITEMS; ++k) {
      if (sel[k]) {
        const uint32_t i = tile * span + k * SUMA_TILE + threadIdx.x;
        const uint32_t dst = s_prefix + rank[k];
        if (dst < SUMA_EXTRACT_CAPACITY && (uint64_t)base + dst < a.arena_cap) {
          Surfel4 s;
          s.a = FLAGS ? sf[4 * (size_t)i] : pa[k];
          s.b = sf[4 * (size_t)i + 1];
          s.c = sf[4 * (size_t)i + 2];
          s.d = sf[4 * (size_t)i + 3];
          store_surfel(a.arena, base + dst, s);
        }
      }
    }
    if (tile == ntiles - 1 && threadIdx.x == 0) a.ds->n_extracted = s_prefix + total;
  }
  if (is_finaliser(s_tile, ntiles)) {
    finalise_tickets(a.ds, a.group_next, a.group_words);
    if (threadIdx.x == 0) commit_extraction(a.ds, a.slots, a.slot, base, a.arena_cap, (ntiles == 0) ? 0u : a.ds->n_extracted);
  }
}

hipError_t launch_extract(suma_ctx* c, uint32_t slot, float cx, float cy, float extent, int use_flags) {
  ExtractArgs a;
  a.flags = use_flags ? c->extract_flags : nullptr;
  a.in = c->surfels[c->cur];
  a.arena = c->cache_arena;
  a.ds = c->ds;
  a.slots = c->cache_slots;
  a.poses = c->poses;
  a.status = c->tile_status;
  a.epoch = ++c->epoch;
  a.group_words = c->group_words;
  a.group = c->tile_group + (size_t)(a.epoch & 1u) * c->group_words;
  a.group_next = c->tile_group + (size_t)((a.epoch + 1u) & 1u) * c->group_words;
  a.slot = slot;
  a.arena_cap = c->cache_cap;
  a.cx = cx;
  a.cy = cy;
  a.extent = extent;
  /* algorithmic bytes (K12 reads the map it filters): 64 S without the selection bytes, S with them */
  ProfScope ps(c, "k12_extract_submap", (use_flags ? 1.0 : 64.0) * (double)c->known_surfels);
  const uint32_t grid = compact_grid(c, ((uint64_t)c->known_surfels + 2 * c->P + K12_ITEMS - 1) / K12_ITEMS);
  if (use_flags)
    k12_extract<true><<<grid, SUMA_TILE, 0, c->ls>>>(a);
  else
    k12_extract<false><<<grid, SUMA_TILE, 0, c->ls>>>(a);
  return hipGetLastError();
}

/* append the cached surfels of one tile to the active map (SurfelMap.cpp:775-780, 801-806); when the update has
 * flagged a tile for extraction, the appended records get their selection byte here (the K12 predicate,
 * extract_surfels.vert:46-64), so that K12 can rely on the bytes for the whole map */
__global__ void __launch_bounds__(256)
    k_append_cached(suma_surfel* surfels, const suma_surfel* arena, DevState* ds, const CacheSlot* slots,
                    uint32_t slot, uint32_t max_surfels, const float* poses, uint8_t* ex_flags, float ex_cx, float ex_cy,
                    float ex_extent) {
  const CacheSlot cs = slots[slot];
  const uint32_t S = ds->n_surfels;
  const float4* __restrict__ src = reinterpret_cast<const float4*>(arena + cs.offset);
  float4* __restrict__ dst = reinterpret_cast<float4*>(surfels);
  uint32_t n = cs.count;
  if ((uint64_t)S + n > max_surfels) n = max_surfels - S;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < 4ull * n; k += (uint64_t)gridDim.x * blockDim.x)
    dst[4ull * S + k] = src[k];
  if (ex_flags != nullptr)
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
      const float4 pa = src[4 * k];
      float Ps[16];
      load_pose(poses, (int32_t)src[4 * k + 2].w, Ps);
      const v3 pos = m4_point(Ps, xyz(pa));
      ex_flags[S + k] = !(sdm_abs(pos.x - ex_cx) > ex_extent || sdm_abs(pos.y - ex_cy) > ex_extent) ? 1 : 0;
    }
}
__global__ void k_append_commit(DevState* ds, const CacheSlot* slots, uint32_t slot, uint32_t max_surfels) {
  uint32_t S = ds->n_surfels, n = slots[slot].count;
  if ((uint64_t)S + n > max_surfels) {
    n = max_surfels - S;
    atomicOr(&ds->overflow, 1u);
  }
  ds->n_surfels = S + n;
}

hipError_t launch_append_cached(suma_ctx* c, uint32_t slot) {
  float ex[3] = {0.f, 0.f, 0.f};
  if (c->flagged.valid) {
    ex[0] = (float)(2.0 * c->flagged.i * c->p.submap_extent); /* submapIndex2center, SurfelMap.cpp:704-706 */
    ex[1] = (float)(2.0 * c->flagged.j * c->p.submap_extent);
    ex[2] = c->p.submap_extent;
  }
  k_append_cached<<<1024, 256, 0, c->ls>>>(c->surfels[c->cur], c->cache_arena, c->ds, c->cache_slots, slot,
                                               c->p.max_surfels, c->poses, c->flagged.valid ? c->extract_flags : nullptr,
                                               ex[0], ex[1], ex[2]);
  k_append_commit<<<1, 1, 0, c->ls>>>(c->ds, c->cache_slots, slot, c->p.max_surfels);
  return hipGetLastError();
}
