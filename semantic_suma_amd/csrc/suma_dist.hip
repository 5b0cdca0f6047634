/*
 * suma_dist.hip -- libsuma_hip_dist.so: the one collective of the multi-GPU path, callable from a C++ host.
 *
 * The path shards over independent units only: one hypothesis or one sequence per GPU, one process
 * per GPU, and per scan ONE all-gather of the ranks' results (16 doubles of pose + a few statistics, ~150 bytes) so
 * that every rank can pick the same winner -- the pattern of the reference's loop-closure verification, which
 * tries several initialisations in sequence on one GPU (SurfelMapping.cpp:662-779).  The collective is latency
 * bound; it runs on the ctx stream behind the minimisation whose result it carries.
 *
 * Kept in its own library so that libsuma_hip.so does not depend on RCCL: single-GPU integrations never load it.
 * Python hosts use torch.distributed (backend "nccl" = RCCL) for the same gather (semantic_suma_amd/distributed.py).
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <string>

#include "../../include/suma_hip_dist.h"

#define SUMA_DIST_MAX_DOUBLES 64u
/* the exchange of suma_run_hypotheses moves an n_hyp x 18 table (n_hyp <= 64): one collective per scan */
#define SUMA_DIST_TABLE_DOUBLES (64u * 18u)

struct suma_dist_comm {
  ncclComm_t comm;
  int world, rank;
  double *d_send, *d_recv; /* device staging */
  double* h_buf;           /* pinned: send block followed by world receive blocks */
  /* all-reduce of the hypothesis table: a stream of the communicator's own, on the communicator's device (round-3
   * advisor: a thread_local stream was created on whatever device was current and never destroyed) */
  int device;
  hipStream_t ar_stream;
  double *d_table, *h_table; /* SUMA_DIST_TABLE_DOUBLES each (pinned host) */
  std::string err;
};

static thread_local std::string g_dist_error;

extern "C" const char* suma_dist_last_error(const suma_dist_comm* c) { return c ? c->err.c_str() : g_dist_error.c_str(); }

extern "C" int suma_dist_unique_id(char id[SUMA_DIST_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= SUMA_DIST_ID_BYTES, "ncclUniqueId does not fit");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) {
    g_dist_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
    return SUMA_ERR_HIP;
  }
  memset(id, 0, SUMA_DIST_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return SUMA_OK;
}

extern "C" int suma_dist_comm_create(const char id[SUMA_DIST_ID_BYTES], int world, int rank, suma_dist_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return SUMA_ERR_INVALID;
  *out = nullptr;
  suma_dist_comm* c = new (std::nothrow) suma_dist_comm();
  if (!c) return SUMA_ERR_NOMEM;
  c->comm = nullptr;
  c->world = world;
  c->rank = rank;
  c->d_send = c->d_recv = c->h_buf = nullptr;
  c->ar_stream = nullptr;
  c->d_table = c->h_table = nullptr;
  c->device = 0;
  (void)hipGetDevice(&c->device); /* the device the communicator is created on (the calling thread's current device) */
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  (void)hipGetLastError(); /* RCCL reports a stale, already handled HIP error of the calling thread as its own */
  ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank); /* on the calling thread's current HIP device */
  if (r != ncclSuccess) {
    g_dist_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
    delete c;
    return SUMA_ERR_HIP;
  }
  const size_t blk = SUMA_DIST_MAX_DOUBLES * sizeof(double);
  if (hipMalloc((void**)&c->d_send, blk) != hipSuccess || hipMalloc((void**)&c->d_recv, blk * (size_t)world) != hipSuccess ||
      hipHostMalloc((void**)&c->h_buf, blk * (size_t)(world + 1), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&c->d_table, SUMA_DIST_TABLE_DOUBLES * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_table, SUMA_DIST_TABLE_DOUBLES * sizeof(double), hipHostMallocDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&c->ar_stream, hipStreamNonBlocking) != hipSuccess) {
    g_dist_error = "suma_dist_comm_create: staging allocation failed";
    suma_dist_comm_destroy(c);
    return SUMA_ERR_HIP;
  }
  *out = c;
  return SUMA_OK;
}

extern "C" void suma_dist_comm_destroy(suma_dist_comm* c) {
  if (!c) return;
  if (c->d_send) hipFree(c->d_send);
  if (c->d_recv) hipFree(c->d_recv);
  if (c->h_buf) hipHostFree(c->h_buf);
  if (c->d_table) hipFree(c->d_table);
  if (c->h_table) hipHostFree(c->h_table);
  if (c->ar_stream) {
    hipStreamSynchronize(c->ar_stream);
    hipStreamDestroy(c->ar_stream);
  }
  if (c->comm) ncclCommDestroy(c->comm);
  delete c;
}

extern "C" int suma_gather(suma_ctx* ctx, suma_dist_comm* c, const double* send, uint32_t count, double* all) {
  if (!ctx || !c || !send || !all || count == 0 || count > SUMA_DIST_MAX_DOUBLES) return SUMA_ERR_INVALID;
  hipStream_t stream = (hipStream_t)suma_ctx_stream(ctx);
  const size_t bytes = (size_t)count * sizeof(double);
  memcpy(c->h_buf, send, bytes);
#define DTRY(expr)                                                      \
  do {                                                                  \
    hipError_t e__ = (expr);                                            \
    if (e__ != hipSuccess) {                                            \
      c->err = std::string(#expr) + ": " + hipGetErrorString(e__);      \
      return SUMA_ERR_HIP;                                              \
    }                                                                   \
  } while (0)
  DTRY(hipMemcpyAsync(c->d_send, c->h_buf, bytes, hipMemcpyHostToDevice, stream));
  ncclResult_t r = ncclAllGather(c->d_send, c->d_recv, count, ncclDouble, c->comm, stream);
  if (r != ncclSuccess) {
    c->err = std::string("ncclAllGather: ") + ncclGetErrorString(r);
    return SUMA_ERR_HIP;
  }
  double* h_recv = c->h_buf + SUMA_DIST_MAX_DOUBLES;
  DTRY(hipMemcpyAsync(h_recv, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost, stream));
  DTRY(hipStreamSynchronize(stream));
#undef DTRY
  memcpy(all, h_recv, bytes * (size_t)c->world);
  return SUMA_OK;
}

extern "C" int suma_gather_poses(suma_ctx* ctx, suma_dist_comm* c, const double pose[16], double* all_poses) {
  return suma_gather(ctx, c, pose, 16, all_poses);
}

/* Element-wise sum over the ranks of a table of doubles: the exchange step of suma_run_hypotheses (include/suma_runner.h;
 * every row of the n_hyp x 18 table is owned by exactly one rank and zero elsewhere, so the sum IS the gathered table).
 * ncclAllReduce on a stream of its own -- the runner's pipeline is internal to it -- in chunks of the staging block. */
static int allreduce_sum_on_device(suma_dist_comm* c, const double* send, uint32_t count, double* out) {
  hipStream_t stream = c->ar_stream;
  /* in place on the device, one collective per SUMA_DIST_TABLE_DOUBLES: the whole n_hyp x 18 table of a scan in one */
  for (uint32_t lo = 0; lo < count; lo += SUMA_DIST_TABLE_DOUBLES) {
    const uint32_t n = count - lo < SUMA_DIST_TABLE_DOUBLES ? count - lo : SUMA_DIST_TABLE_DOUBLES;
    memcpy(c->h_table, send + lo, n * sizeof(double));
    if (hipMemcpyAsync(c->d_table, c->h_table, n * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) {
      c->err = "suma_dist_allreduce_sum: upload failed";
      return SUMA_ERR_HIP;
    }
    ncclResult_t r = ncclAllReduce(c->d_table, c->d_table, n, ncclDouble, ncclSum, c->comm, stream);
    if (r != ncclSuccess) {
      c->err = std::string("ncclAllReduce: ") + ncclGetErrorString(r);
      return SUMA_ERR_HIP;
    }
    if (hipMemcpyAsync(c->h_table, c->d_table, n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
      c->err = "suma_dist_allreduce_sum: download failed";
      return SUMA_ERR_HIP;
    }
    memcpy(out + lo, c->h_table, n * sizeof(double));
  }
  return SUMA_OK;
}
extern "C" int suma_dist_allreduce_sum(suma_dist_comm* c, const double* send, uint32_t count, double* out) {
  if (!c || !send || !out || count == 0) return SUMA_ERR_INVALID;
  /* the communicator's device is made current for the call and the caller's device is put back on every path: the
   * calling thread belongs to the embedding application (round-4 advisor) */
  int cur = -1;
  const bool switched = hipGetDevice(&cur) == hipSuccess && cur != c->device;
  if (switched && hipSetDevice(c->device) != hipSuccess) {
    c->err = "suma_dist_allreduce_sum: cannot make the communicator's device current";
    return SUMA_ERR_HIP;
  }
  const int r = allreduce_sum_on_device(c, send, count, out);
  if (switched) hipSetDevice(cur);
  return r;
}
/* the same with the signature of suma_exchange_fn: pass it to suma_run_hypotheses with user = the suma_dist_comm */
extern "C" int suma_dist_exchange(void* user, const double* local, double* all, uint32_t n_doubles) {
  return suma_dist_allreduce_sum((suma_dist_comm*)user, local, n_doubles, all);
}
