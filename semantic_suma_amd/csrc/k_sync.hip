/*
 * k_sync.hip -- in-memory hand-offs between the two HIP streams of a scan pipeline.
 *
 * The scan pipeline keeps work that is not on a scan's critical path (the preprocessing K1-K3 of the NEXT scan) on a
 * side stream, where it overlaps the surfel passes of the current scan; the reference does everything on the one
 * thread / GL context it has (SurfelMapping.cpp:175-210).  The consumer must not start before the producer is done.
 * A runtime event dependency (hipEventRecord + hipStreamWaitEvent) between two streams stalls the waiting stream for
 * ~10 us on this platform even when the event is long complete (tools/xstream.hip), which is as much as the work it
 * would hide.  Instead: the producer stream ends its batch with k_signal, which stores a sequence number; the
 * consumer stream runs k_gate -- one wave that polls the word with agent-scope loads and s_sleep -- in front of its
 * dependent kernels.  The kernel boundary behind the gate gives the usual visibility of the producer's writes (they
 * were written back at the producer kernels' end, before k_signal ran).  The spin is bounded: a producer that never
 * arrives surfaces as DevState.overflow bit 3 (SUMA_ERR_HIP), never as a hung GPU.
 */
#include <atomic>
#include <cstddef>
#include <cstdlib>

#include "suma_internal.h"

__global__ void k_signal(uint32_t* word, uint32_t seq) {
  __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_gate(const uint32_t* word, uint32_t seq, uint32_t* fault) {
  if (threadIdx.x != 0) return;
  for (uint32_t spins = 0; spins < (1u << 24); ++spins) {
    /* sequence numbers only grow; compare modulo 2^32 */
    if ((int32_t)(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0) return;
    __builtin_amdgcn_s_sleep(16);
  }
  atomicOr(fault, 8u);
  atomicOr(fault + (offsetof(DevState, fault_site) - offsetof(DevState, overflow)) / 4, 0x1u); /* the stream gate */
}

hipError_t launch_signal(suma_ctx* c, hipStream_t st, uint32_t word, uint32_t seq) {
  k_signal<<<1, 1, 0, st>>>(c->sync_flags + word, seq);
  return hipGetLastError();
}
hipError_t launch_gate(suma_ctx* c, hipStream_t st, uint32_t word, uint32_t seq) {
  k_gate<<<1, 64, 0, st>>>(c->sync_flags + word, seq, &c->ds->overflow);
  return hipGetLastError();
}
/* Contexts in this process that own a side stream.  The in-memory gate is a kernel that occupies its hardware queue
 * while it polls: with ONE pipeline the producer always has a queue of its own, with several (suma_run_sequences: up to
 * six pipelines x three streams on GPU_MAX_HW_QUEUES queues, four by default) pipeline A's gate can sit in front of
 * pipeline B's producer while B's gate sits in front of A's -- both spin until the bounded time-out (round-3 review).
 * So: more than one such context alive (or SUMA_GATE_EVENTS=1) -> the hand-off is a runtime event dependency
 * (hipEventRecord + hipStreamWaitEvent), which costs the waiting stream ~10 us but cannot deadlock; the other
 * pipelines fill that bubble. */
static std::atomic<int> g_side_ctxs(0);

int ensure_side_stream(suma_ctx* c) {
  if (c->side_stream || c->side_stream_off) return SUMA_OK;
  /* A tool that SERIALISES kernel execution breaks the in-memory hand-off (the gate runs alone, the signal never
   * starts): rocprofv3 --pmc does (it exports ROCPROF_COUNTER_COLLECTION), and so does AMD_SERIALIZE_KERNEL.  Under
   * either, and under SUMA_NO_SIDE_STREAM=1 (A/B measurements), everything stays on the ctx stream. */
  const char* ser = getenv("AMD_SERIALIZE_KERNEL");
  const bool serialised = getenv("ROCPROF_COUNTER_COLLECTION") != nullptr || (ser && atoi(ser) != 0);
  if (getenv("SUMA_NO_SIDE_STREAM") || serialised) {
    c->side_stream_off = 1;
    return SUMA_OK;
  }
  HIP_TRY(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
  g_side_ctxs.fetch_add(1); /* counted with the stream: side_stream_released() decrements whenever the stream exists */
  HIP_TRY(c, hipEventCreateWithFlags(&c->pre_event, hipEventDisableTiming));
  HIP_TRY(c, hipEventCreateWithFlags(&c->order_event, hipEventDisableTiming));
  HIP_TRY(c, hipMalloc((void**)&c->zbuf_k1, c->P * 8));
  HIP_TRY(c, hipMemsetAsync(c->zbuf_k1, 0xFF, c->P * 8, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SUMA_OK;
}
void side_stream_released(suma_ctx* c) {
  if (c->side_stream) g_side_ctxs.fetch_sub(1);
}

int side_handoff(suma_ctx* c, const suma_frame* frame) {
  static const bool force_events = getenv("SUMA_GATE_EVENTS") != nullptr;
  c->pre_seq += 1;
  if (force_events || g_side_ctxs.load() > 1) {
    HIP_TRY(c, hipEventRecord(c->pre_event, c->side_stream));
    c->gate_by_event = 1;
  } else {
    HIP_TRY(c, launch_signal(c, c->side_stream, 0, c->pre_seq));
    c->gate_by_event = 0;
  }
  c->gate_pending = c->pre_seq; /* the first reader of the frame on the ctx stream issues the wait (flush_gate) */
  c->gate_frame = frame;
  return SUMA_OK;
}

hipError_t flush_gate(suma_ctx* c) {
  if (!c->gate_pending) return hipSuccess;
  const uint32_t seq = c->gate_pending;
  c->gate_pending = 0;
  c->gate_frame = nullptr;
  if (c->gate_by_event) return hipStreamWaitEvent(c->stream, c->pre_event, 0);
  return launch_gate(c, c->stream, 0, seq);
}
