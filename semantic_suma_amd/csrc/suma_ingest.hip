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
 */
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>

#include "suma_internal.h"

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

struct Ingest {
  suma_pipeline* s;
  hipStream_t copy_stream;
  IngestSlot slot[INGEST_SLOTS];
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
  hipSetDevice(g->s->c->device);
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

static int ingest_get(suma_pipeline* s, Ingest** out) {
  if (s->ingest) {
    *out = s->ingest;
    return SUMA_OK;
  }
  suma_ctx* c = s->c;
  Ingest* g = new (std::nothrow) Ingest();
  if (!g) return SUMA_ERR_NOMEM;
  g->s = s;
  g->head = g->tail = 0;
  g->stop = false;
  for (auto& q : g->slot) {
    memset(&q, 0, sizeof(q));
  }
  HIP_TRY(c, hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
  for (auto& q : g->slot) {
    HIP_TRY(c, hipEventCreateWithFlags(&q.uploaded, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&q.consumed, hipEventDisableTiming));
  }
  g->worker = std::thread(ingest_main, g);
  s->ingest = g;
  *out = g;
  return SUMA_OK;
}

void ingest_destroy(suma_pipeline* s) {
  Ingest* g = s ? s->ingest : nullptr;
  if (!g) return;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->stop = true;
  }
  g->cv.notify_all();
  if (g->worker.joinable()) g->worker.join();
  hipStreamSynchronize(g->copy_stream);
  for (auto& q : g->slot) {
    if (q.pinned) hipHostFree(q.pinned);
    if (q.device) hipFree(q.device);
    if (q.uploaded) hipEventDestroy(q.uploaded);
    if (q.consumed) hipEventDestroy(q.consumed);
  }
  hipStreamDestroy(g->copy_stream);
  delete g;
  s->ingest = nullptr;
}

extern "C" int suma_pipeline_prefetch_scan(suma_pipeline* s, const suma_float4* points, const float* labels,
                                           const float* probs, uint32_t n) {
  if (!s || (n > 0 && !points)) return SUMA_ERR_INVALID;
  Ingest* g = nullptr;
  int r = ingest_get(s, &g);
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

extern "C" int suma_pipeline_process_prefetched(suma_pipeline* s, int32_t fixed_iterations) {
  if (!s || !s->ingest) return SUMA_ERR_INVALID;
  Ingest* g = s->ingest;
  suma_ctx* c = s->c;
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
  /* the stream that runs the scan's preprocessing waits for the upload (device-side dependency) */
  int r = pipeline_process_scan_impl(s, (const suma_float4*)q->device,
                                     q->labels ? (const float*)(q->device + labels_offset(n)) : nullptr,
                                     q->probs ? (const float*)(q->device + probs_offset(n)) : nullptr, n,
                                     fixed_iterations, q->uploaded);
  const hipError_t ec = hipEventRecord(q->consumed, c->stream);
  {
    std::lock_guard<std::mutex> lk(g->mu);
    q->consumed_valid = (ec == hipSuccess);
    q->state = 0;
    g->head += 1;
  }
  return r;
}

extern "C" int suma_pipeline_process_scan_async(suma_pipeline* s, const suma_float4* points, const float* labels,
                                                const float* probs, uint32_t n, int32_t fixed_iterations) {
  if (!s || (n > 0 && !points)) return SUMA_ERR_INVALID;
  bool staged = false;
  if (s->ingest) {
    Ingest* g = s->ingest;
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
