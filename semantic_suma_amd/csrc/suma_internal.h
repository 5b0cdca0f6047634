/*
 * suma_internal.h -- host-side state of the gfx950 core and the launcher prototypes shared by
 * the translation units (k_preprocess.hip, k_icp.hip, k_render.hip, k_update.hip, suma_api.hip).
 */
#ifndef SUMA_INTERNAL_H_
#define SUMA_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/suma_hip.h"
#include "dev_math.h"

#ifndef SUMA_TILE
#define SUMA_TILE 1024u       /* items per compaction tile = threads per block (16 waves) */
#endif
#ifndef SUMA_COMPACT_BLOCKS
#define SUMA_COMPACT_BLOCKS 256u /* grid of the ticketed compaction kernels: ONE block per CU -- their 1024-thread blocks are
                                    resident one per CU (registers / LDS), every block draws tickets until none is left, so
                                    a second generation of blocks only starts to draw failing tickets at the kernel's
                                    tail (512: K9 76 -> 74 us slower, -0.7 % scans/s) */
#endif
#define SUMA_STREAM_BLOCKS 2048u /* grid cap of the grid-stride surfel kernels: 8 blocks of 256 per CU */
#define SUMA_EXTRACT_CAPACITY 500000u /* SurfelMap.cpp:279 */
#define SUMA_MAX_MODEL_WIDTH 5461u /* floor((2^21 - 1) / (1.5 * 256)): k_render's window coordinates */
#define SUMA_MAX_HYP 64u

/* Counters that live in HBM so that no kernel launch needs a host round trip. */
struct DevState {
  uint32_t n_surfels;      /* S: size of the active map */
  uint32_t n_updated;      /* S': survivors of K9 (before the K11 area filter) */
  uint32_t n_kept_updated; /* survivors of K9 and K11 */
  uint32_t n_data;         /* D: new surfels emitted by K10 (before the K11 area filter) */
  uint32_t n_kept_data;
  uint32_t n_extracted;    /* K12: surfels written by the last extraction */
  uint32_t overflow;       /* bit 0: surfel capacity, bit 1: cache arena, bit 2: extract capacity, bit 3: compaction spin limit */
  uint32_t ticket;         /* dynamic tile id of the look-back compaction kernels */
  uint32_t reserved0;      /* blocks-done counter of a self-closing objective pass (k_icp.hip) */
  uint32_t cache_used;     /* surfels allocated from the submap cache arena */
  uint32_t fault_site;     /* which bounded spin gave up (bit per site; reported with overflow bit 3) */
  uint32_t n_ext_update;   /* extraction fused into the update: records K9 sent to the cache arena (K10's come behind) */
  uint32_t pad[4];
};

/* one cached submap tile in the device arena */
struct CacheSlot {
  uint32_t offset, count;
};

/* Gauss-Newton state of one minimisation, device resident (LieGaussNewton members,
 * LieGaussNewton.h:56-72, plus Frame2Model's iteration counter and row 7 of its blend target) */
struct GnState {
  double Tk[16];
  double last_error;
  double F, F_inlier;
  uint32_t iteration; /* Frame2Model::iteration_ */
  uint32_t k;         /* LieGaussNewton::k_ */
  uint32_t done, converged;
  uint32_t valid, outlier, invalid, n_hist;
  uint32_t pending; /* a pixel phase has left partial sums that the next launch must consume */
  uint32_t pad[3];
  int64_t acc[SUMA_ACC_WORDS];
  double JtJ[36];
  double Jtr[6];
};

/* What the host reads back after a minimisation of the scan pipeline.  The closing launch of the chain
 * writes it straight into pinned host memory (seq last, behind a system-scope fence), so the result
 * costs no copy commands in the stream: the re-rendering that follows the minimisation starts right
 * behind the closing launch, and the host polls seq. */
struct HostResult {
  double Tk[16];
  double F, F_inlier;
  uint32_t valid, outlier, invalid, k, converged, iteration;
  DevState ds;
  uint32_t seq;
  uint32_t n_hist;
  /* only filled for the class-by-class entries (suma_icp_minimize / suma_icp_jacobian_products), not by the scan
   * pipeline's launches: the fixed-point sums of the last step with the bias removed (JtJ / Jtr / F are these words
   * times 2^-28, formed on the host exactly as the device forms them) */
  int64_t acc[SUMA_ACC_WORDS];
  double JtJ[36], Jtr[6]; /* of the last step (LieGaussNewton::information); closing launch of suma_icp_minimize only */
};

struct MapConsts {
  float pixel_size, log_prior, log_unstable, p_unstable;
  float radconf_angle_thresh, update_angle_thresh;
};

struct suma_frame {
  suma_ctx* ctx;
  uint32_t width, height;
  float4* map[3]; /* vertex, normal, semantic: one allocation */
  /* bumped by every call of the C-ABI that writes the frame (upload / copy / swap / preprocess / render / touch): what
   * the render de-duplication and the fused K8 products compare instead of assuming that a caller-owned frame changed */
  uint64_t version;
  /* ctx-stream accesses (suma_ctx.enq_seq) -- lets suma_preprocess put its side-stream work in front of everything
   * the ctx stream still holds, unless that includes an access to this very frame */
  uint64_t last_access;
};

struct ProfEvent {
  hipEvent_t a, b;
  hipStream_t stream; /* the launch stream the pair brackets */
  int id;
  double bytes;
  uint32_t launches; /* kernel launches bracketed by this event pair (a chain of identical launches) */
};

/* host-side time of the blocking host-vector entry (suma_pipeline_process_scan), summed since the last reset:
 * where a call of the reference-shaped entry spends its time on the CALLER's thread (suma_pipeline_host_entry_times) */
struct HostEntryTimes {
  double slot_wait_s;  /* waiting for the staging slot of the scan before last to be read */
  double copy_s;       /* pageable -> pinned copies (caller + helper threads) */
  double enqueue_s;    /* hipMemcpyAsync + event record of the upload */
  double launch_s;     /* enqueueing the scan's kernels (everything else that is not a wait) */
  double result_wait_s; /* polling for the minimisation result: the GPU is the one being waited for */
  double call_s;       /* whole calls */
  uint64_t calls;
  uint32_t copy_threads; /* caller + helpers that took a share of the copies */
};

struct suma_ctx {
  suma_params p;
  int device;
  hipStream_t stream;      /* the ctx stream: everything the C-ABI promises to order */
  hipStream_t ls;          /* stream the launchers enqueue on: == stream, except while the scan pipeline enqueues side work */
  hipStream_t side_stream; /* work that is off the critical path of a scan (the next scan's upload + preprocessing) */
  int side_stream_off;     /* SUMA_NO_SIDE_STREAM / a serialising tool: everything on the ctx stream */
  uint32_t* sync_flags;    /* device: sequence words of the in-memory stream hand-offs (k_sync.hip) */
  uint32_t pre_seq;        /* preprocessing hand-offs issued so far */
  uint32_t gate_pending;   /* != 0: the ctx stream has not yet waited for this preprocessing hand-off (flush_gate) */
  int gate_by_event;       /* the pending hand-off is a runtime event (pre_event), not the in-memory word: several
                              pipelines in one process, see side_handoff (k_sync.hip) */
  hipEvent_t pre_event;
  hipEvent_t order_event;  /* ctx stream -> side stream, only when a frame's last access may still be in flight */
  uint64_t enq_seq, done_seq; /* frame accesses enqueued on the ctx stream / known complete at the last host wait */
  bool cache_nothing_stale; /* the last compaction attempt found no stale block: no point in synchronising again */
  const suma_frame* gate_frame; /* the frame the pending side-stream work writes */
  struct Ingest* ingest;   /* pinned double-buffered scan staging + copy stream + helper threads (suma_ingest.hip) */
  HostResult* h_rec;       /* pinned: results of suma_icp_minimize / suma_icp_jacobian_products ([0] / [1]) */
  uint32_t rec_seq;
  int gn_host_full;        /* the reporting launch being enqueued also writes HostResult.acc / n_hist */
  std::string err;

  proj_t pd, pm; /* data / model projection */
  MapConsts mc;
  size_t P, Pm;

  /* preprocessing scratch */
  unsigned long long* zbuf_data; /* P keys: K7 (and K1 outside the scan pipeline's side stream) */
  unsigned long long* zbuf_k1;   /* P keys: K1 of the scan pipeline -- preprocessing of scan t+1 overlaps K7 / K10 of scan t */
  float4* eroded;                /* P: raw labels of K1 (scratch between k1_resolve and the fused K2/K3) */
  /* optional vertex-map filters (k_filters.hip), allocated on first use */
  float4* filt_temp;             /* P: the reference's temp_vertices_ */
  unsigned long long* filt_sort; /* 2 x filt_cap keys (pixel << 32 | point index), unsorted / sorted */
  void* filt_sort_tmp;
  size_t filt_sort_tmp_bytes;
  uint32_t filt_cap;
  float4* scan_points;           /* staging for host scans */
  float *scan_labels, *scan_probs;
  uint32_t scan_cap;

  /* ICP */
  const suma_frame *icp_current, *icp_model;
  suma_icp_objective obj; /* per-object Frame2Model parameters of the adapter (suma_icp_set_objective) */
  bool obj_set;
  GnState* gn;        /* 2 x SUMA_MAX_HYP states, alternating with the launch parity */
  int64_t* gn_partial; /* 3 rotating sets of SUMA_MAX_HYP x ICP_RECORDS x SUMA_ACC_WORDS accumulators (k_icp.hip) */
  uint32_t gn_part_launch;   /* rotation counter, never reset */
  uint32_t gn_part_dirty[3]; /* hypotheses with possibly non-zero records, per set */
  uint32_t gn_launch;  /* launches since the last gn_init */
  /* per-pixel K8 products already written for (frame, stamp) by the statistics pass, see launch_map_update */
  const suma_frame* k8_fused_frame;
  uint64_t k8_fused_version; /* suma_frame.version the products were made from */
  uint32_t k8_fused_stamp;
  uint64_t k8_fused_params;
  int gn_fuse_k8; /* the next eval-only pixel launch also runs K8's per-pixel work and the counter resets */
  HostResult* gn_fused_report; /* if set: the next eval-only pixel launch closes itself and reports here (gn_host_seq) */
  HostResult* gn_host_out; /* if set: the closing launch being enqueued reports to this pinned host record ... */
  uint32_t gn_host_seq;    /* ... and stamps it with this sequence number */
  int gn_emit_pose;        /* the closing launch of the chain being enqueued writes pose_block */
  double gn_pose_base[16];
  float* pose_block;       /* device: 16 floats pose + 16 floats inverse for the post-ICP render */
  int gn_init_pending; /* the next k_icp_iter launch starts a fresh single chain from gn_T0_host */
  uint32_t gn_iteration0;
  HostEntryTimes het;
  uint32_t icp_iteration0; /* suma_icp_set_iteration: Frame2Model::iteration_ for the NEXT suma_icp_minimize (one shot) */
  double gn_T0_host[16];
  double* gn_history;  /* (max_iterations + 1) x 16 doubles (single minimise only) */
  double* gn_T0s;      /* SUMA_MAX_HYP x 16 staging for batched starts */
  uint32_t gn_history_cap;
  uint32_t last_n_hist; /* LieGaussNewton::history() entries of the last suma_icp_minimize */
  uint64_t hist_seq;    /* minimisations that recorded a history on this context (suma_icp_history_sequence) */
  uint32_t icp_blocks;
  GnState* h_gn; /* pinned */

  /* surfel map */
  suma_surfel* surfels[2]; /* double buffer: active map / compaction target */
  int cur;
  float* poses;     /* max_poses x 16 */
  float* poses_inv; /* max_poses x 16 */
  suma_frame *old_frame, *new_frame, *composed_frame;
  unsigned long long *zbuf_a, *zbuf_b; /* Pm */
  float4* radius_conf;                 /* P */
  float4* pixrec;                      /* P x 4: packed measurement record for K9 (one 64-byte line per pixel) */
  uint8_t* integrated;                 /* P */
  /* one byte per surfel of the compaction target: "lies in the submap tile that is extracted right after this update"
   * (written by K9 / K10 at the surfel's final index, read by K12 instead of a pass over the whole map) */
  uint8_t* extract_flags;              /* max_surfels */
  struct {
    bool valid;                        /* the last update flagged the tile (i, j) */
    int32_t i, j;
    bool fused;    /* ... and K9 / K10 have already written and committed the tile's cache block (slot below) */
    uint32_t slot;
  } flagged;
  uint32_t* index_map;                 /* P: K7 winners as surfel id + 1 (exported by K10) */
  unsigned long long* tile_status;     /* look-back status words */
  unsigned long long* tile_group;      /* 2 x group_words, per 64 tiles: {arrived, sum}; launches alternate halves */
  uint32_t group_words;
  uint32_t n_tiles_cap;
  uint32_t epoch;
  DevState* ds;
  DevState* h_ds; /* pinned */
  uint32_t timestamp; /* SurfelMap::timestamp_ (host copy; kernels get it by value) */
  /* submaps (SurfelMap.cpp:744-824): caches live in a device arena, the index on the host */
  int32_t origin_i, origin_j;
  suma_surfel* cache_arena;
  uint32_t cache_cap;
  CacheSlot* cache_slots; /* device table */
  uint32_t cache_compactions; /* times the arena has been compacted (cache_compact, suma_api.hip) */
  uint64_t cache_bound;       /* host-side upper bound of DevState.cache_used: exact value at the last read-back + the
                                 most every extraction since can have added */
  uint32_t cache_slots_cap;
  std::map<std::pair<int32_t, int32_t>, uint32_t> cache_index; /* (i,j) -> slot */
  std::vector<std::pair<int32_t, int32_t>> extraction;        /* pending tiles, used as a stack */

  /* profiling */
  int profiling; /* 0 off, 1 every kernel group, 2 only the group named prof_filter, 3 every 4th occurrence of that group */
  uint32_t prof_tick;
  std::string prof_filter;
  std::vector<ProfEvent> prof_events;
  std::vector<hipEvent_t> prof_pool;
  std::vector<std::string> prof_names;
  std::vector<double> prof_ms, prof_bytes;
  std::vector<uint64_t> prof_launches;
  uint32_t known_surfels; /* last S read back (for the algorithmic-byte model) */

  /* what the OLD / NEW frames and the last render() target currently hold: lets render() skip a
   * call that would reproduce them bit for bit (the reference renders the same map from the same
   * pose at the end of scan t and again at the start of scan t+1, SurfelMapping.cpp:351,803) */
  uint64_t map_version, params_version;
  /* K7 splat already in zbuf_data (fused into the post-ICP render pass of the pipeline) */
  struct {
    bool valid;
    float pose[16];
    uint64_t map_version, params_version;
  } k7;
  /* suma_map_render_active splats the index map speculatively (the reference's updatePose renders the active map at
   * the pose that updateMap then passes to update(), SurfelMapping.cpp:406 / :799); a splat nobody consumed switches the
   * speculation off until an update arrives that WOULD have consumed one */
  struct {
    bool on;
    bool have_last;
    float last_pose[16];
    uint64_t map_version, params_version;
  } k7_spec;
  struct {
    bool valid;
    float pose_old[16], pose_new[16], conf_threshold;
    uint64_t map_version, params_version;
    const suma_frame* out;
    uint64_t out_version, old_version, new_version; /* suma_frame.version of the three targets after the render */
  } rendered;
};

struct suma_pipeline {
  suma_ctx* c;
  suma_frame *last_frame, *current_frame, *current_model, *last_model;
  double current_pose[16], last_pose[16], pose_old[16], pose_new[16], last_increment[16];
  double last_pose_old[16]; /* lastPose_old_, SurfelMapping.cpp:456 */
  uint32_t timestamp;
  int phase; /* 0: between scans, 1: begin_scan done, 2: update_pose done (suma_pipeline_begin_scan / _update_pose / _update_map) */
  float log_unstable;
  suma_icp_stats stats;
  uint32_t track_loss;
  /* the statistics pass of updatePose (SurfelMapping.cpp:411-423) is read back lazily: its copy is
   * enqueued, and resolved at the next synchronisation point instead of stalling the scan */
  HostResult* h_res; /* pinned: [0] minimisation result, [1..2] statistics pass (alternating) */
  uint32_t res_seq, stats_seq;
  bool stats_pending;
  uint32_t stats_slot;
  suma_icp_stats stats_mst;
};

int pipeline_process_scan_impl(suma_pipeline* s, const suma_float4* d_points, const float* d_labels, const float* d_probs,
                               uint32_t n, int32_t fixed_iterations, hipEvent_t upload_done);
int pipeline_begin_scan_impl(suma_pipeline* s, const suma_float4* d_points, const float* d_labels, const float* d_probs,
                             uint32_t n, hipEvent_t upload_done);
int pipeline_update_pose_impl(suma_pipeline* s, int32_t fixed_iterations);
int pipeline_update_map_impl(suma_pipeline* s);
/* the stream behind which a scan's input buffers are free again (the preprocessing that read them runs there) */
hipStream_t pipeline_input_stream(suma_pipeline* s);
/* suma_ingest.hip */
void ingest_destroy(suma_ctx* c);
void ingest_drain(suma_ctx* c);
/* host scan -> pinned staging -> copy stream; *d_base = device block (points | labels | probs), *uploaded = the event the
 * consumer stream waits for, *slot = token for ingest_consumed */
int ingest_stage_blocking(suma_ctx* c, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                          const suma_float4** d_points, const float** d_labels, const float** d_probs, hipEvent_t* uploaded,
                          void** slot);
void ingest_consumed(suma_ctx* c, void* slot, hipStream_t reader);
/* side stream of a ctx (created on first use; NULL under SUMA_NO_SIDE_STREAM / serialising tools) */
int ensure_side_stream(suma_ctx* c);
/* after side-stream work that writes `frame`: the ctx stream's next reader waits for it (flush_gate) */
int side_handoff(suma_ctx* c, const suma_frame* frame);
void side_stream_released(suma_ctx* c);
int pipeline_process_host_scan(suma_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, int32_t fixed_iterations);

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                           \
      return SUMA_ERR_HIP;                                                                       \
    }                                                                                            \
  } while (0)

/* profiling scope: brackets the launches of one named kernel with events on the ctx stream */
int prof_begin(suma_ctx* c, const char* name, double bytes, uint32_t launches);
void prof_end(suma_ctx* c, int token);
struct ProfScope {
  suma_ctx* c;
  int tok;
  ProfScope(suma_ctx* c_, const char* name, double bytes, uint32_t launches = 1)
      : c(c_), tok(wanted(c_, name) ? prof_begin(c_, name, bytes, launches) : -1) {}
  static bool wanted(suma_ctx* c, const char* name) {
    if (c->profiling == 1) return true;
    if (c->profiling < 2 || c->prof_filter != name) return false;
    /* an event record is a barrier + signal packet: ~6 us of bubble in front of the kernel behind it (rocprofv3
     * timeline).  Mode 3 samples one occurrence in four so that the measurement costs the measured run < 1 % */
    return c->profiling == 2 || (c->prof_tick++ & 3u) == 0;
  }
  ~ProfScope() {
    if (tok >= 0) prof_end(c, tok);
  }
};

/* ---- launchers (each enqueues on c->stream, returns hipGetLastError()) ---- */
/* k_preprocess.hip */
hipError_t launch_preprocess(suma_ctx* c, const float4* d_pts, const float* d_labels, const float* d_probs, uint32_t n,
                             uint32_t timestamp, suma_frame* out);
/* k_icp.hip */
hipError_t launch_k1_average(suma_ctx* c, const float4* d_pts, const float* d_labels, const float* d_probs, uint32_t n,
                             uint32_t timestamp, float4* vertex, float4* raw_semantic);
hipError_t launch_k1c_bilateral(suma_ctx* c, float4* vertex);
hipError_t launch_gn_init(suma_ctx* c, const double* h_T0s, uint32_t n_hyp, int with_history, uint32_t iteration0,
                          uint32_t iteration0_rest = 0);
hipError_t launch_icp_iteration(suma_ctx* c, uint32_t n_hyp, uint32_t max_iter, double epsilon, double delta,
                                int eval_only, int with_history, int pixel);
const GnState* gn_result(suma_ctx* c);
/* k_render.hip */
hipError_t launch_map_render(suma_ctx* c, const float* pose_old, const float* pose_new, float conf_threshold,
                             suma_frame* out);
hipError_t launch_map_render_single(suma_ctx* c, const float* pose, float conf_threshold, int active, int fuse_k7,
                                    suma_frame* mirror);
hipError_t launch_map_render_composed(suma_ctx* c, const float* pose_old, const float* pose_new,
                                      float conf_threshold);
/* k_update.hip */
/* ex: optional centre (x, y) + half-width of the submap tile that will be extracted right after this update;
 * fused_slot >= 0: the update performs that extraction itself, into this slot of the cache table */
hipError_t launch_map_update(suma_ctx* c, const float* pose, const float* inv_pose, const suma_frame* f, float cx,
                             float cy, float extent, int k7_done, const float* ex, int fused_slot);
hipError_t launch_clear_index_zbuf(suma_ctx* c);
K8Out launch_k8_out(suma_ctx* c);
hipError_t launch_set_poses(suma_ctx* c, const float* d_src, uint32_t first, uint32_t n);
hipError_t launch_fill_identity_poses(suma_ctx* c);
hipError_t launch_extract(suma_ctx* c, uint32_t slot, float cx, float cy, float extent, int use_flags);
hipError_t launch_append_cached(suma_ctx* c, uint32_t slot);

/* k_sync.hip: in-memory hand-offs between the ctx stream and the side stream.  A runtime event dependency between two
 * HIP streams costs ~10 us of stall on this platform (tools/xstream.hip: ping-pong 42 us vs 22 us for the same two
 * 10 us kernels on one stream); a one-wave gate kernel that polls a sequence word costs ~2 us. */
hipError_t launch_signal(suma_ctx* c, hipStream_t st, uint32_t word, uint32_t seq);
hipError_t launch_gate(suma_ctx* c, hipStream_t st, uint32_t word, uint32_t seq);
/* makes the ctx stream wait for the pending preprocessing hand-off (one-wave gate kernel) */
hipError_t flush_gate(suma_ctx* c);

/* host helper shared by api + pipeline */
void rigid_inverse_f(const float* m, float* out);

#endif
