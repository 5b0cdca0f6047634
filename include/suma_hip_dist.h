/*
 * suma_hip_dist.h -- C-ABI of libsuma_hip_dist.so: the multi-GPU side of the MI355X projective-ICP core.
 *
 * One process per GPU (one suma_ctx each).  The path shards only over independent units -- one ICP hypothesis or one
 * sequence per GPU (the reference runs its hypotheses one after the other on one GPU,
 * SurfelMapping.cpp:662-779) -- and needs exactly one exchange step: an all-gather of each rank's result (pose +
 * statistics, <= 64 doubles) over RCCL / xGMI, after which every rank selects the same winner.  Single-GPU hosts do
 * not need this library (libsuma_hip.so does not link RCCL).
 *
 * Bootstrap as with NCCL: rank 0 calls suma_dist_unique_id and distributes the 128 bytes over whatever channel
 * the host application has (MPI, a file, a socket); every rank then calls suma_dist_comm_create on the thread whose
 * current HIP device is its ctx's device.
 */
#ifndef SUMA_HIP_DIST_H_
#define SUMA_HIP_DIST_H_

#include "suma_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SUMA_DIST_ID_BYTES 128

typedef struct suma_dist_comm suma_dist_comm;

int suma_dist_unique_id(char id[SUMA_DIST_ID_BYTES]);
int suma_dist_comm_create(const char id[SUMA_DIST_ID_BYTES], int world, int rank, suma_dist_comm** out);
void suma_dist_comm_destroy(suma_dist_comm* comm);
const char* suma_dist_last_error(const suma_dist_comm* comm);

/* all-gather of `count` (<= 64) doubles per rank on the ctx stream, i.e. behind everything enqueued on the ctx so
 * far; all = world x count doubles in rank order.  Blocks until the result is on the host. */
int suma_gather(suma_ctx* ctx, suma_dist_comm* comm, const double* send, uint32_t count, double* all);
/* the gather of SURVEY.md 8(b): each rank's 4x4 pose (column-major doubles); all_poses = world x 16 doubles */
int suma_gather_poses(suma_ctx* ctx, suma_dist_comm* comm, const double pose[16], double* all_poses);

/* element-wise sum over the ranks of `count` doubles (any count), on a stream of the library's own; blocks until the
 * result is on the host.  suma_dist_exchange has the signature of suma_exchange_fn (include/suma_runner.h): BASELINE
 * configs[2] from a C++ host is  suma_run_hypotheses(&params, device, &job, 0, suma_dist_exchange, comm, poses, winners, err). */
int suma_dist_allreduce_sum(suma_dist_comm* comm, const double* send, uint32_t count, double* out);
int suma_dist_exchange(void* comm, const double* local, double* all, uint32_t n_doubles);

#ifdef __cplusplus
}
#endif
#endif
