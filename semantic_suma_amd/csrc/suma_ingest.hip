/*
 * suma_ingest.hip -- device-side scan ingest for the scan pipeline.
 *
 * Replaces the hand-over of the reference, where KITTIReader::read (src/io/KITTIReader.cpp:136-203) fills an
 * rv::Laserscan in pageable host memory and SurfelMapping::processScan (SurfelMapping.cpp:175-210, :323-331) assigns
 * it to GL buffers with a blocking glBufferData on the thread that owns the GL context.
 *
 * Here: three staging slots, each a pinned host block + a device block, a copy stream and one ingest thread.
 *   suma_pipeline_prefetch_scan       hands (points, labels, probs, n) to the ingest thread and returns at once;
 *                                     the thread copies the arrays into the slot's pinned block, enqueues the
 *                                     H2D copies on the copy stream and records the slot's event;
 *   suma_pipeline_process_prefetched  makes the compute stream wait on that event (device-side dependency, the
 *                                     host does not block on the copy) and runs the scan from the device block.
 * With prefetch(k+1) (and k+2) issued before process(k), the host copy and the PCIe transfer of the next scans
 * overlap the kernels of scan k: staging one scan (a ~3 MB host copy into pinned memory, ~250 us on one core, plus
 * ~60 us of DMA) takes about as long as processing one, so it needs a head start of two scans to stay hidden.  A slot is refilled (scan k+2) only after the `consumed` event recorded on the compute stream
 * behind scan k has completed: the upload that read the pinned block and the preprocessing kernels that read the
 * device block are both behind it.
 *
 * The plain host-pointer entry suma_pipeline_process_scan (pipeline_process_host_scan below) uses the same pieces on
 * the caller's time line: the scan goes into one of two pinned blocks of its own -- copied by the caller and
 * COPY_HELPERS helper threads, one core alone needs about a scan's worth of GPU time for 3 MB -- and up through the
 * copy stream; the call returns as early as the device-pointer entry, so the next call's copy overlaps this scan's
 * surfel passes.
 */
#include <string.h>

#include <time.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "suma_internal.h"

#include <sched.h>
#include <stdio.h>

#define INGEST_SLOTS 3u

struct IngestSlot {
  /* request */
  const suma_float4* points;
  const float *labels, *probs;
  uint32_t n;
  /* staging */
  char* pinned;
  char* device;
  size_t cap_bytes;
  hipEvent_t uploaded;
  hipEvent_t consumed; /* recorded on the compute stream behind the scan that read this slot */
  bool consumed_valid;
  int state; /* 0 free, 1 requested, 2 staged (copies enqueued, event recorded), -1 failed */
  hipError_t error;
};

/* The blocking host-pointer entry copies a scan (~3 MB) into pinned memory on the CALLER's time line: one core moves
 * that in ~250 us, about a whole scan's worth of GPU time.  A few helper threads that sleep on a condition variable
 * between scans take a share each. */
#define HOST_CHUNKS 4 /* pieces of a host scan whose pinned copy and DMA are pipelined (blocking entry) */
#define COPY_HELPERS 7
#define COPY_SEGMENTS 3 /* points, labels, probs */
struct CopySeg {
  char* dst;
  const char* src;
  size_t bytes;
};
struct CopyPool {
  int helpers; /* helper threads that exist (0 .. COPY_HELPERS): sized from the CPUs this process may use */
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> th;
  CopySeg job[COPY_HELPERS][COPY_SEGMENTS];
  /* A helper that went to sleep on the condition variable needs 30-60 us to run again, as long as the whole copy takes;
   * scans arrive back to back, so a helper first SPINS on the generation word for SPIN_NS after its last job (and
   * costs nothing once the caller stops sending scans: it then sleeps until the next post). */
  std::atomic<uint64_t> posted;
  std::atomic<int> pending;
  std::atomic<int> sleepers;
  std::atomic<bool> stop;
};
#define SPIN_NS 400000ll

static long long mono_ns() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

static void copy_helper(CopyPool* p, int id) {
  uint64_t seen = 0;
  long long idle_since = mono_ns();
  for (;;) {
    uint64_t g = p->posted.load(std::memory_order_acquire);
    if (g == seen) {
      if (p->stop.load(std::memory_order_relaxed)) return;
      if (mono_ns() - idle_since < SPIN_NS) {
        __builtin_ia32_pause();
        continue;
      }
      std::unique_lock<std::mutex> lk(p->mu);
      p->sleepers.fetch_add(1);
      p->cv.wait(lk, [&] { return p->stop.load() || p->posted.load(std::memory_order_acquire) != seen; });
      p->sleepers.fetch_sub(1);
      if (p->stop.load()) return;
      continue;
    }
    seen = g;
    for (int k = 0; k < COPY_SEGMENTS; ++k) {
      const CopySeg j = p->job[id][k];
      if (j.bytes) memcpy(j.dst, j.src, j.bytes);
    }
    p->pending.fetch_sub(1, std::memory_order_release);
    idle_since = mono_ns();
  }
}

/* share `part` (0 = the caller) of a segment split into helpers + 1 page-aligned shares */
static CopySeg seg_share(const CopySeg& s, int part, int helpers) {
  const size_t parts = (size_t)helpers + 1;
  const size_t share = ((s.bytes / parts) + 4095) & ~(size_t)4095;
  const size_t lo = share * (size_t)part;
  if (s.bytes == 0 || lo >= s.bytes) return {nullptr, nullptr, 0};
  const size_t n = (part == (int)parts - 1 || lo + share > s.bytes) ? s.bytes - lo : share;
  return {s.dst + lo, s.src + lo, n};
}

/* every segment dst <- src, split over the helpers and the calling thread */
static void pool_copy(CopyPool* p, const CopySeg* segs, int nseg) {
  size_t total = 0;
  for (int k = 0; k < nseg; ++k) total += segs[k].bytes;
  if (total < (128u << 10) || p->th.empty()) {
    for (int k = 0; k < nseg; ++k)
      if (segs[k].bytes) memcpy(segs[k].dst, segs[k].src, segs[k].bytes);
    return;
  }
  const int H = p->helpers;
  for (int h = 0; h < H; ++h)
    for (int k = 0; k < COPY_SEGMENTS; ++k) p->job[h][k] = (k < nseg) ? seg_share(segs[k], h + 1, H) : CopySeg{nullptr, nullptr, 0};
  p->pending.store(H, std::memory_order_relaxed);
  p->posted.fetch_add(1, std::memory_order_release);
  if (p->sleepers.load() > 0) {
    std::lock_guard<std::mutex> lk(p->mu); /* pairs with the predicate check of a helper about to sleep */
    p->cv.notify_all();
  }
  for (int k = 0; k < nseg; ++k) {
    const CopySeg mine = seg_share(segs[k], 0, H);
    if (mine.bytes) memcpy(mine.dst, mine.src, mine.bytes);
  }
  while (p->pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
}

struct Ingest {
  suma_ctx* c;
  hipStream_t copy_stream;
  IngestSlot slot[INGEST_SLOTS];
  /* blocking entry (suma_pipeline_process_scan): two staging slots of its own, used alternately */
  IngestSlot bslot[2];
  uint32_t bnext;
  CopyPool pool;
  uint32_t head, tail; /* next slot to process / next slot to fill (counts, slot = count % INGEST_SLOTS) */
  std::mutex mu;
  std::condition_variable cv;
  std::thread worker;
  bool stop;
};

static size_t scan_bytes(uint32_t n) {
  /* points, then labels, then probs; each block 256-byte aligned */
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  return up((size_t)n * sizeof(float4)) + 2 * up((size_t)n * sizeof(float));
}
static size_t labels_offset(uint32_t n) { return (((size_t)n * sizeof(float4)) + 255) & ~(size_t)255; }
static size_t probs_offset(uint32_t n) { return labels_offset(n) + ((((size_t)n * sizeof(float)) + 255) & ~(size_t)255); }

static hipError_t slot_reserve(Ingest* g, IngestSlot* q, uint32_t n) {
  const size_t need = scan_bytes(n ? n : 1);
  if (need <= q->cap_bytes) return hipSuccess;
  const size_t cap = need + need / 4;
  if (q->pinned) hipHostFree(q->pinned);
  if (q->device) hipFree(q->device);
  q->pinned = q->device = nullptr;
  q->cap_bytes = 0;
  hipError_t e = hipHostMalloc((void**)&q->pinned, cap, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&q->device, cap);
  if (e != hipSuccess) return e;
  q->cap_bytes = cap;
  return hipSuccess;
}

static void ingest_main(Ingest* g) {
  hipSetDevice(g->c->device);
  uint32_t next = 0; /* requests are served in order */
  for (;;) {
    IngestSlot* q;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv.wait(lk, [&] { return g->stop || (next != g->tail && g->slot[next % INGEST_SLOTS].state == 1); });
      if (g->stop) return;
      q = &g->slot[next % INGEST_SLOTS];
    }
    hipError_t e = hipSuccess;
    if (q->consumed_valid) e = hipEventSynchronize(q->consumed); /* previous user of this slot has finished */
    if (e == hipSuccess) e = slot_reserve(g, q, q->n);
    const uint32_t n = q->n;
    if (e == hipSuccess && n > 0) {
      memcpy(q->pinned, q->points, (size_t)n * sizeof(float4));
      size_t bytes = (size_t)n * sizeof(float4);
      if (q->labels) {
        memcpy(q->pinned + labels_offset(n), q->labels, (size_t)n * sizeof(float));
        bytes = labels_offset(n) + (size_t)n * sizeof(float);
      }
      if (q->probs) {
        memcpy(q->pinned + probs_offset(n), q->probs, (size_t)n * sizeof(float));
        bytes = probs_offset(n) + (size_t)n * sizeof(float);
      }
      /* one transfer: the gaps between the blocks are at most 2 x 255 bytes */
      e = hipMemcpyAsync(q->device, q->pinned, bytes, hipMemcpyHostToDevice, g->copy_stream);
    }
    if (e == hipSuccess) e = hipEventRecord(q->uploaded, g->copy_stream);
    {
      std::lock_guard<std::mutex> lk(g->mu);
      q->error = e;
      q->state = (e == hipSuccess) ? 2 : -1;
    }
    g->cv.notify_all();
    ++next;
  }
}

/* CPUs this process may use: the smaller of the affinity mask and the cgroup v2 quota (cpu.max = "quota period") */
static int usable_cpus() {
  int n = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    double period = 0.0;
    if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) {
      const int quota = (int)(atof(q) / period + 0.5);
      if (quota >= 1 && quota < n) n = quota;
    }
    fclose(f);
  }
  return n > 0 ? n : 1;
}
static int ingest_default_helpers() {
  const int n = usable_cpus();
  /* Half of what is left beside the caller (the other half stays with the application's own threads), three at most:
   * 8 CPUs and more -> 3 helpers, 4 -> 1, 2 -> 0.  Measured on the pool's box under taskset
   * (profiles/r06_host_entry_cpus.jsonl): the 3 MB of a 64 x 2048 scan take 88 us on the caller alone, 65 with one
   * helper, 62 with three, 50 with seven -- every helper beyond the third buys 3 us and spins a CPU for it. */
  const int h = (n - 1) / 2;
  return h > 3 ? 3 : h;
}

static int ingest_get(suma_ctx* c, Ingest** out) {
  if (c->ingest) {
    *out = c->ingest;
    return SUMA_OK;
  }
  Ingest* g = new (std::nothrow) Ingest();
  if (!g) return SUMA_ERR_NOMEM;
  g->c = c;
  g->head = g->tail = 0;
  g->stop = false;
  for (auto& q : g->slot) {
    memset(&q, 0, sizeof(q));
  }
  for (auto& q : g->bslot) memset(&q, 0, sizeof(q));
  g->bnext = 0;
  g->pool.posted = 0;
  g->pool.pending = 0;
  g->pool.sleepers = 0;
  g->pool.stop = false;
  HIP_TRY(c, hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
  for (auto& q : g->slot) {
    HIP_TRY(c, hipEventCreateWithFlags(&q.uploaded, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&q.consumed, hipEventDisableTiming));
  }
  for (auto& q : g->bslot) {
    HIP_TRY(c, hipEventCreateWithFlags(&q.uploaded, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&q.consumed, hipEventDisableTiming));
  }
  /* Helper threads for the pageable -> pinned copy, sized from the CPUs this process may really use: the affinity mask
   * AND the cgroup's CPU quota (a GPU box of the pool shows 256 cores and grants 16; the round-5 driver run lost 11 % of
   * the host-vector rate with eight copying threads next to pytest's own).  A helper SPINS between scans (SPIN_NS), so
   * each one is a CPU taken: one CPU stays with the caller, one with the rest of the process, at most COPY_HELPERS
   * help.  SUMA_COPY_HELPERS=n overrides (0: the caller copies alone). */
  int helpers = ingest_default_helpers();
  const char* nh = getenv("SUMA_COPY_HELPERS");
  if (nh) helpers = atoi(nh);
  if (helpers < 0) helpers = 0;
  if (helpers > COPY_HELPERS) helpers = COPY_HELPERS;
  g->pool.helpers = helpers;
  for (int h = 0; h < helpers; ++h) g->pool.th.emplace_back(copy_helper, &g->pool, h);
  c->het.copy_threads = (uint32_t)helpers + 1u;
  g->worker = std::thread(ingest_main, g);
  c->ingest = g;
  *out = g;
  return SUMA_OK;
}

void ingest_destroy(suma_ctx* c) {
  Ingest* g = c ? c->ingest : nullptr;
  if (!g) return;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->stop = true;
  }
  g->cv.notify_all();
  if (g->worker.joinable()) g->worker.join();
  {
    std::lock_guard<std::mutex> lk(g->pool.mu);
    g->pool.stop = true;
  }
  g->pool.cv.notify_all();
  for (auto& t : g->pool.th)
    if (t.joinable()) t.join();
  hipStreamSynchronize(g->copy_stream);
  for (IngestSlot* q = g->slot; q != g->slot + INGEST_SLOTS; ++q) {
    if (q->pinned) hipHostFree(q->pinned);
    if (q->device) hipFree(q->device);
    if (q->uploaded) hipEventDestroy(q->uploaded);
    if (q->consumed) hipEventDestroy(q->consumed);
  }
  for (auto& q : g->bslot) {
    if (q.pinned) hipHostFree(q.pinned);
    if (q.device) hipFree(q.device);
    if (q.uploaded) hipEventDestroy(q.uploaded);
    if (q.consumed) hipEventDestroy(q.consumed);
  }
  hipStreamDestroy(g->copy_stream);
  delete g;
  c->ingest = nullptr;
}

/* scans staged ahead of their turn that nobody will process (a failed sequence, suma_pipeline_reset): wait for their
 * uploads and mark the slots free, so that the next sequence on this pipeline does not consume them (round-3 advisor) */
void ingest_drain(suma_ctx* c) {
  Ingest* g = c ? c->ingest : nullptr;
  if (!g) return;
  std::unique_lock<std::mutex> lk(g->mu);
  while (g->head != g->tail) {
    IngestSlot* q = &g->slot[g->head % INGEST_SLOTS];
    g->cv.wait(lk, [&] { return q->state == 2 || q->state == -1; });
    q->state = 0;
    g->head += 1;
  }
  lk.unlock();
  hipStreamSynchronize(g->copy_stream);
  for (auto& q : g->slot) q.consumed_valid = false;
}

extern "C" int suma_pipeline_prefetch_scan(suma_pipeline* s, const suma_float4* points, const float* labels,
                                           const float* probs, uint32_t n) {
  if (!s || (n > 0 && !points)) return SUMA_ERR_INVALID;
  Ingest* g = nullptr;
  int r = ingest_get(s->c, &g);
  if (r) return r;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->tail - g->head >= INGEST_SLOTS) {
    s->c->err = "suma_pipeline_prefetch_scan: all staging slots hold scans that have not been processed";
    return SUMA_ERR_INVALID;
  }
  IngestSlot* q = &g->slot[g->tail % INGEST_SLOTS];
  q->points = points;
  q->labels = labels;
  q->probs = probs;
  q->n = n;
  q->state = 1;
  g->tail += 1;
  g->cv.notify_all();
  return SUMA_OK;
}

/* the oldest staged scan: begin (phases_only) or the whole scan */
static int run_prefetched(suma_pipeline* s, int32_t fixed_iterations, bool begin_only) {
  if (!s || !s->c->ingest) return SUMA_ERR_INVALID;
  Ingest* g = s->c->ingest;
  suma_ctx* c = s->c;
  /* the phase is checked BEFORE a slot is taken: a call in the wrong phase must not drop a staged scan */
  if (s->phase != 0) {
    c->err = "suma_pipeline_process_prefetched: the previous scan has not been closed with suma_pipeline_update_map";
    return SUMA_ERR_INVALID;
  }
  IngestSlot* q;
  {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->head == g->tail) {
      c->err = "suma_pipeline_process_prefetched: no scan has been prefetched";
      return SUMA_ERR_INVALID;
    }
    q = &g->slot[g->head % INGEST_SLOTS];
    g->cv.wait(lk, [&] { return q->state == 2 || q->state == -1; }); /* copies enqueued (not: completed) */
    if (q->state == -1) {
      c->err = std::string("scan staging failed: ") + hipGetErrorString(q->error);
      q->state = 0;
      g->head += 1;
      return SUMA_ERR_HIP;
    }
  }
  const uint32_t n = q->n;
  const suma_float4* dp = (const suma_float4*)q->device;
  const float* dl = q->labels ? (const float*)(q->device + labels_offset(n)) : nullptr;
  const float* dq = q->probs ? (const float*)(q->device + probs_offset(n)) : nullptr;
  /* the stream that runs the scan's preprocessing waits for the upload (device-side dependency) */
  int r = pipeline_begin_scan_impl(s, dp, dl, dq, n, q->uploaded);
  /* the slot is free again once the preprocessing that read its device block has run: recorded on the stream that
   * carries it (the side stream, where an event record costs the scan nothing) */
  const hipError_t ec = hipEventRecord(q->consumed, pipeline_input_stream(s));
  {
    std::lock_guard<std::mutex> lk(g->mu);
    q->consumed_valid = (ec == hipSuccess);
    q->state = 0;
    g->head += 1;
  }
  if (r == SUMA_OK && !begin_only) {
    r = pipeline_update_pose_impl(s, fixed_iterations);
    if (r == SUMA_OK) r = pipeline_update_map_impl(s);
  }
  if (r != SUMA_OK) s->phase = 0;
  return r;
}
extern "C" int suma_pipeline_process_prefetched(suma_pipeline* s, int32_t fixed_iterations) {
  return run_prefetched(s, fixed_iterations, false);
}
extern "C" int suma_pipeline_begin_prefetched(suma_pipeline* s) { return run_prefetched(s, 0, true); }

extern "C" int suma_pipeline_process_scan_async(suma_pipeline* s, const suma_float4* points, const float* labels,
                                                const float* probs, uint32_t n, int32_t fixed_iterations) {
  if (!s || (n > 0 && !points)) return SUMA_ERR_INVALID;
  bool staged = false;
  if (s->c->ingest) {
    Ingest* g = s->c->ingest;
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->head != g->tail) {
      const IngestSlot& q = g->slot[g->head % INGEST_SLOTS];
      staged = (q.points == points && q.labels == labels && q.probs == probs && q.n == n);
      if (!staged) {
        s->c->err = "suma_pipeline_process_scan_async: a different scan is staged ahead of this one";
        return SUMA_ERR_INVALID;
      }
    }
  }
  if (!staged) {
    int r = suma_pipeline_prefetch_scan(s, points, labels, probs, n);
    if (r) return r;
  }
  return suma_pipeline_process_prefetched(s, fixed_iterations);
}

/* A pageable host scan on its way to the device, on the CALLER's time line: the scan is copied into one of two pinned
 * blocks by the caller and the helper threads and uploaded on the copy stream, both in HOST_CHUNKS pieces -- the DMA of
 * piece k runs while piece k + 1 is being copied, so the scan is on the device about one piece after its last byte
 * was copied.  Returns the device pointers and the event the consumer stream must wait for; the caller hands the slot
 * back with ingest_consumed once the reader has been enqueued. */
int ingest_stage_blocking(suma_ctx* c, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                          const suma_float4** d_points, const float** d_labels, const float** d_probs, hipEvent_t* uploaded,
                          void** slot) {
  Ingest* g = nullptr;
  int r = ingest_get(c, &g);
  if (r) return r;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->head != g->tail) {
      c->err = "host scan while prefetched scans are waiting (suma_pipeline_process_prefetched comes first)";
      return SUMA_ERR_INVALID;
    }
  }
  IngestSlot* q = &g->bslot[g->bnext++ & 1u];
  long long t0 = mono_ns();
  if (q->consumed_valid) HIP_TRY(c, hipEventSynchronize(q->consumed)); /* the scan before last has read this slot */
  HIP_TRY(c, slot_reserve(g, q, n));
  long long t1 = mono_ns();
  c->het.slot_wait_s += 1e-9 * (double)(t1 - t0);
  if (n > 0) {
    size_t bytes = (size_t)n * sizeof(float4);
    if (labels) bytes = labels_offset(n) + (size_t)n * sizeof(float);
    if (probs) bytes = probs_offset(n) + (size_t)n * sizeof(float);
    const size_t piece = ((bytes / HOST_CHUNKS) + 4095) & ~(size_t)4095;
    /* the staging layout is points | labels | probs at fixed offsets: walk it piece by piece, copying from whichever
     * source arrays overlap the piece */
    const size_t off_l = labels_offset(n), off_p = probs_offset(n);
    const struct { size_t off, len; const char* src; } part[3] = {
        {0, (size_t)n * sizeof(float4), (const char*)points},
        {off_l, labels ? (size_t)n * sizeof(float) : 0, (const char*)labels},
        {off_p, probs ? (size_t)n * sizeof(float) : 0, (const char*)probs}};
    for (size_t lo = 0; lo < bytes; lo += piece) {
      const size_t hi = lo + piece < bytes ? lo + piece : bytes;
      CopySeg ps[COPY_SEGMENTS];
      int np = 0;
      for (int k = 0; k < 3; ++k) {
        const size_t a = part[k].off > lo ? part[k].off : lo;
        const size_t b = (part[k].off + part[k].len) < hi ? (part[k].off + part[k].len) : hi;
        if (part[k].len && a < b) ps[np++] = {q->pinned + a, part[k].src + (a - part[k].off), b - a};
      }
      pool_copy(&g->pool, ps, np);
      const long long t2 = mono_ns();
      c->het.copy_s += 1e-9 * (double)(t2 - t1);
      HIP_TRY(c, hipMemcpyAsync(q->device + lo, q->pinned + lo, hi - lo, hipMemcpyHostToDevice, g->copy_stream));
      t1 = mono_ns();
      c->het.enqueue_s += 1e-9 * (double)(t1 - t2);
    }
  }
  HIP_TRY(c, hipEventRecord(q->uploaded, g->copy_stream));
  c->het.enqueue_s += 1e-9 * (double)(mono_ns() - t1);
  *d_points = (const suma_float4*)q->device;
  *d_labels = labels ? (const float*)(q->device + labels_offset(n)) : nullptr;
  *d_probs = probs ? (const float*)(q->device + probs_offset(n)) : nullptr;
  *uploaded = q->uploaded;
  *slot = (void*)q;
  return SUMA_OK;
}
/* where the calls of the blocking host-vector entry spent their time on the caller's thread, as sums since the last
 * reset: out = {calls, call, slot wait, copy, upload enqueue, kernel enqueue, result wait, copy threads} (seconds) */
extern "C" int suma_pipeline_host_entry_times(suma_pipeline* s, double out[8], int reset) {
  if (!s || !out) return SUMA_ERR_INVALID;
  HostEntryTimes& h = s->c->het;
  out[0] = (double)h.calls;
  out[1] = h.call_s;
  out[2] = h.slot_wait_s;
  out[3] = h.copy_s;
  out[4] = h.enqueue_s;
  out[5] = h.launch_s;
  out[6] = h.result_wait_s;
  out[7] = (double)h.copy_threads;
  if (reset) {
    const uint32_t t = h.copy_threads;
    memset(&h, 0, sizeof(h));
    h.copy_threads = t;
  }
  return SUMA_OK;
}

void ingest_consumed(suma_ctx* c, void* slot, hipStream_t reader) {
  IngestSlot* q = (IngestSlot*)slot;
  (void)c;
  q->consumed_valid = (hipEventRecord(q->consumed, reader) == hipSuccess);
}

/* SurfelMapping::processScan with host vectors, as the reference's caller hands them over (SurfelMapping.cpp:175-210):
 * staged as above, the preprocessing waits for the upload on the device -- the call returns as early as the
 * device-pointer entry does, so the surfel passes of this scan overlap the next call's copy. */
static int run_host_scan(suma_pipeline* s, const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                         int32_t fixed_iterations, bool begin_only) {
  suma_ctx* c = s->c;
  const suma_float4* dp;
  const float *dl, *dq;
  hipEvent_t up;
  void* slot;
  const long long t_call = mono_ns();
  const double waits_before = c->het.slot_wait_s + c->het.copy_s + c->het.enqueue_s + c->het.result_wait_s;
  int r = ingest_stage_blocking(c, points, labels, probs, n, &dp, &dl, &dq, &up, &slot);
  if (r) return r;
  r = pipeline_begin_scan_impl(s, dp, dl, dq, n, up);
  ingest_consumed(c, slot, pipeline_input_stream(s));
  if (r == SUMA_OK && !begin_only) {
    r = pipeline_update_pose_impl(s, fixed_iterations);
    if (r == SUMA_OK) r = pipeline_update_map_impl(s);
  }
  if (r != SUMA_OK) s->phase = 0;
  const double dt = 1e-9 * (double)(mono_ns() - t_call);
  c->het.call_s += dt;
  c->het.calls += 1;
  c->het.launch_s += dt - ((c->het.slot_wait_s + c->het.copy_s + c->het.enqueue_s + c->het.result_wait_s) - waits_before);
  return r;
}
int pipeline_process_host_scan(suma_pipeline* s, const suma_float4* points, const float* labels, const float* probs,
                               uint32_t n, int32_t fixed_iterations) {
  return run_host_scan(s, points, labels, probs, n, fixed_iterations, false);
}
/* initialize + preprocess of SurfelMapping::processScan with the host vectors the reference's caller holds */
extern "C" int suma_pipeline_begin_scan(suma_pipeline* s, const suma_float4* points, const float* labels,
                                        const float* probs, uint32_t n) {
  if (!s || (n > 0 && !points)) return SUMA_ERR_INVALID;
  return run_host_scan(s, points, labels, probs, n, 0, true);
}
