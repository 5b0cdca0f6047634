/*
 * suma_runner.h -- native host loops for the replica configurations (part of libsuma_hip.so).
 *
 * The reference drives one sequence from one thread (visualizer loop -> SurfelMapping::processScan,
 * SurfelMapping.cpp:175-210).  One MI355X has room for several such sequences at once -- a single sequence is a
 * chain of ~20 dependent kernels that keeps a fraction of the chip busy -- and BASELINE.json asks for two replica
 * workloads: "all 11 KITTI odometry sequences concurrently" (configs[3]) and "8 ICP pose hypotheses per scan"
 * (configs[2]; the reference's own multi-start pattern is SurfelMapping.cpp:662-779).  These entry points run those
 * loops in C++: one std::thread per concurrent pipeline, no interpreter between two scans.
 */
#ifndef SUMA_RUNNER_H_
#define SUMA_RUNNER_H_

#include "suma_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one scan: rv::Laserscan's arrays (points n x (x, y, z, 1), labels_float, labels_prob; the last two may be NULL) */
typedef struct suma_scan_ref {
  const suma_float4* points;
  const float* labels;
  const float* probs;
  uint32_t n;
} suma_scan_ref;

/* one sequence = one SurfelMapping object fed scan after scan */
typedef struct suma_sequence_job {
  const suma_scan_ref* scans;
  uint32_t n_scans;
  int32_t on_device; /* != 0: the arrays are device addresses (scans resident in HBM); else host arrays, staged
                        through the pipeline's ingest (pinned slots, copy stream) two scans ahead */
} suma_sequence_job;

typedef struct suma_sequence_result {
  int32_t status;      /* SUMA_OK or the error of the scan that failed */
  uint32_t scans_done;
  uint32_t map_surfels;
  uint32_t track_loss; /* scans on which the frame-to-frame fallback ran */
  double end_pose[16]; /* currentPose_ after the last scan, column-major */
  double seconds;      /* wall time of this sequence (pipeline creation excluded) */
  char error[160];
} suma_sequence_result;

/* The reference's caller IS a native loop -- the visualizer thread calls SurfelMapping::processScan scan after scan
 * (VisualizerWindow.cpp:636-689 -> SurfelMapping.cpp:175).  This is that loop on an EXISTING pipeline: the next
 * job->n_scans scans of its sequence, without a reset and without a stream synchronisation at the end (the pipeline
 * stays as asynchronous as after a single suma_pipeline_process_scan* call).  Host arrays take the blocking host-vector
 * entry (no look-ahead), device arrays the resident entry.  *scans_done = scans that went through. */
int suma_pipeline_run_scans(suma_pipeline* pipeline, const suma_sequence_job* job, int32_t fixed_iterations,
                            uint32_t* scans_done, double* seconds_per_call /* NULL, or n_scans host times: a stall of the
                            calling thread or of a copy helper shows up as ONE long call */);

/* BASELINE configs[3]: runs the jobs in the order given (sort them longest first for an LPT schedule), at most
 * max_concurrent at a time, each through a pipeline of its own on hip_device.  Returns SUMA_OK when every job
 * succeeded; results[k] belongs to jobs[k] either way. */
int suma_run_sequences(const suma_params* params, int hip_device, const suma_sequence_job* jobs, uint32_t n_jobs,
                       uint32_t max_concurrent, int32_t fixed_iterations, suma_sequence_result* results);

/* BASELINE configs[2]: per scan n_hyp Gauss-Newton chains from the starts lastIncrement * perturbations[k]; hypothesis
 * k belongs to rank k % world.  A rank minimises its own hypotheses as one batch, hands the n_hyp x 18 result table
 * (16 pose doubles, error, valid pairs; rows of other ranks' hypotheses zero) to `exchange`, which returns the
 * element-wise SUM over all ranks (= the gathered table: every row is owned by exactly one rank); every rank then
 * picks the same winner -- smallest error per valid pair, ties to the lower index -- and updates its map with it, so
 * the maps stay identical without map traffic.  exchange == NULL for world == 1.
 * poses: n_scans x 16 doubles (currentPose_ after each scan); winners: n_scans (-1 for the first scan). */
typedef int (*suma_exchange_fn)(void* user, const double* local, double* all, uint32_t n_doubles);
typedef struct suma_hypothesis_job {
  const suma_scan_ref* scans;
  uint32_t n_scans;
  int32_t on_device;
  const double* perturbations; /* n_hyp x 16 doubles, column-major 4x4 each; [0] = identity for "unperturbed" */
  uint32_t n_hyp, rank, world;
} suma_hypothesis_job;
int suma_run_hypotheses(const suma_params* params, int hip_device, const suma_hypothesis_job* job,
                        int32_t fixed_iterations, suma_exchange_fn exchange, void* user, double* poses,
                        int32_t* winners, char error[160]);

#ifdef __cplusplus
}
#endif
#endif /* SUMA_RUNNER_H_ */
