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
}

hipError_t launch_signal(suma_ctx* c, hipStream_t st, uint32_t word, uint32_t seq) {
  k_signal<<<1, 1, 0, st>>>(c->sync_flags + word, seq);
  return hipGetLastError();
}
hipError_t launch_gate(suma_ctx* c, hipStream_t st, uint32_t word, uint32_t seq) {
  k_gate<<<1, 64, 0, st>>>(c->sync_flags + word, seq, &c->ds->overflow);
  return hipGetLastError();
}
hipError_t flush_gate(suma_ctx* c) {
  if (!c->gate_pending) return hipSuccess;
  const uint32_t seq = c->gate_pending;
  c->gate_pending = 0;
  return launch_gate(c, c->stream, 0, seq);
}
