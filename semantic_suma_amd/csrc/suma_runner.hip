/*
 * suma_runner.hip -- native host loops for the replica configurations (include/suma_runner.h): BASELINE configs[3]
 * (several sequences at once, one pipeline + one host thread each) and configs[2] (several pose hypotheses per scan).
 * Host code only; everything goes through the public C-ABI of suma_hip.h.
 */
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/suma_runner.h"

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void set_error(char* dst, size_t cap, const std::string& msg) {
  if (!dst || !cap) return;
  strncpy(dst, msg.c_str(), cap - 1);
  dst[cap - 1] = 0;
}

/* one sequence through pipeline `s` (SurfelMapping::processScan scan after scan), which is reset first */
static void run_one_sequence(suma_pipeline* s, const suma_sequence_job& job, int32_t fixed_iterations,
                             suma_sequence_result* res) {
  memset(res, 0, sizeof(*res));
  int r = suma_pipeline_reset(s);
  const double t0 = now_s();
  uint32_t k = 0;
  if (r == SUMA_OK && job.on_device) {
    for (; k < job.n_scans && r == SUMA_OK; ++k) {
      const suma_scan_ref& sc = job.scans[k];
      r = suma_pipeline_process_scan_device(s, sc.points, sc.labels, sc.probs, sc.n, fixed_iterations);
    }
  } else if (r == SUMA_OK) {
    /* host arrays: keep two scans staged beyond the one being processed (suma_ingest.hip) */
    uint32_t staged = 0;
    for (; k < job.n_scans && r == SUMA_OK; ++k) {
      while (staged < job.n_scans && staged < k + 3 && r == SUMA_OK) {
        const suma_scan_ref& sc = job.scans[staged];
        r = suma_pipeline_prefetch_scan(s, sc.points, sc.labels, sc.probs, sc.n);
        ++staged;
      }
      if (r == SUMA_OK) r = suma_pipeline_process_prefetched(s, fixed_iterations);
    }
  }
  if (r == SUMA_OK) r = suma_synchronize(suma_pipeline_ctx(s));
  res->seconds = now_s() - t0;
  res->status = r;
  res->scans_done = (r == SUMA_OK) ? k : (k ? k - 1 : 0);
  if (r != SUMA_OK) set_error(res->error, sizeof(res->error), suma_last_error(suma_pipeline_ctx(s)));
  suma_pipeline_pose(s, res->end_pose);
  res->track_loss = suma_pipeline_track_loss(s);
  uint32_t n = 0;
  if (suma_map_size(suma_pipeline_ctx(s), &n) == SUMA_OK) res->map_surfels = n;
}

extern "C" int suma_pipeline_run_scans(suma_pipeline* s, const suma_sequence_job* job, int32_t fixed_iterations,
                                       uint32_t* scans_done, double* seconds_per_call) {
  if (!s || !job || (!job->scans && job->n_scans)) return SUMA_ERR_INVALID;
  int r = SUMA_OK;
  uint32_t k = 0;
  double t_prev = seconds_per_call ? now_s() : 0.0;
  for (; k < job->n_scans && r == SUMA_OK; ++k) {
    const suma_scan_ref& sc = job->scans[k];
    r = job->on_device ? suma_pipeline_process_scan_device(s, sc.points, sc.labels, sc.probs, sc.n, fixed_iterations)
                       : suma_pipeline_process_scan(s, sc.points, sc.labels, sc.probs, sc.n, fixed_iterations);
    if (seconds_per_call) {
      const double t = now_s();
      seconds_per_call[k] = t - t_prev;
      t_prev = t;
    }
  }
  if (scans_done) *scans_done = (r == SUMA_OK) ? k : (k ? k - 1 : 0);
  return r;
}

extern "C" int suma_run_sequences(const suma_params* params, int hip_device, const suma_sequence_job* jobs,
                                  uint32_t n_jobs, uint32_t max_concurrent, int32_t fixed_iterations,
                                  suma_sequence_result* results) {
  if (!params || (!jobs && n_jobs) || (!results && n_jobs)) return SUMA_ERR_INVALID;
  if (n_jobs == 0) return SUMA_OK;
  if (max_concurrent == 0) max_concurrent = 1;
  const uint32_t n_workers = max_concurrent < n_jobs ? max_concurrent : n_jobs;
  std::atomic<uint32_t> next(0);
  /* one pipeline per WORKER, reset between its sequences (SurfelMapping::reset): a pipeline owns several GB of map
   * and cache arena, creating one per sequence would cost more than a short sequence takes */
  auto worker = [&]() {
    suma_pipeline* s = nullptr;
    int rc = (hipSetDevice(hip_device) == hipSuccess) ? suma_pipeline_create(params, hip_device, &s) : SUMA_ERR_HIP;
    for (;;) {
      const uint32_t j = next.fetch_add(1);
      if (j >= n_jobs) break;
      if (rc != SUMA_OK) {
        memset(&results[j], 0, sizeof(results[j]));
        results[j].status = rc;
        set_error(results[j].error, sizeof(results[j].error), std::string("suma_pipeline_create: ") + suma_last_error(nullptr));
        continue;
      }
      run_one_sequence(s, jobs[j], fixed_iterations, &results[j]);
    }
    if (s) suma_pipeline_destroy(s);
  };
  std::vector<std::thread> th;
  for (uint32_t w = 1; w < n_workers; ++w) th.emplace_back(worker);
  worker(); /* the calling thread is one of the workers */
  for (auto& t : th) t.join();
  int rc = SUMA_OK;
  for (uint32_t j = 0; j < n_jobs; ++j)
    if (results[j].status != SUMA_OK && rc == SUMA_OK) rc = results[j].status;
  return rc;
}

/* C = A * B, column-major, in the fixed operation order every rank uses (distributed.py mul4) */
static void mul4(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] = ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}

extern "C" int suma_run_hypotheses(const suma_params* params, int hip_device, const suma_hypothesis_job* job,
                                   int32_t fixed_iterations, suma_exchange_fn exchange, void* user, double* poses,
                                   int32_t* winners, char error[160]) {
  if (!params || !job || !poses || !winners || job->n_hyp == 0 || job->n_hyp > 64 || job->world == 0 ||
      job->rank >= job->world || (job->world > 1 && !exchange) || !job->perturbations)
    return SUMA_ERR_INVALID;
  suma_pipeline* s = nullptr;
  int r = suma_pipeline_create(params, hip_device, &s);
  if (r != SUMA_OK) {
    set_error(error, 160, std::string("suma_pipeline_create: ") + suma_last_error(nullptr));
    return r;
  }
  const uint32_t n_hyp = job->n_hyp;
  std::vector<uint32_t> mine;
  for (uint32_t k = job->rank; k < n_hyp; k += job->world) mine.push_back(k);
  std::vector<double> starts(16 * mine.size()), Ts(16 * mine.size()), local(18 * (size_t)n_hyp), all(18 * (size_t)n_hyp);
  std::vector<suma_icp_stats> stats(mine.size());
  double increment[16];
  for (int i = 0; i < 16; ++i) increment[i] = (i % 5 == 0) ? 1.0 : 0.0;
  /* A rank that fails must still take part in the NEXT exchange, or its peers wait in the collective for ever (round-3
   * advisor): it stops working on its pipeline, sends a table whose every row carries NaN in the residual column, every
   * rank sees the NaN in the summed table and all of them leave at the same scan. */
  int failed = SUMA_OK;
  auto fail_local = [&](int code) {
    if (failed == SUMA_OK) {
      failed = code;
      set_error(error, 160, suma_last_error(suma_pipeline_ctx(s)));
    }
  };
  for (uint32_t t = 0; t < job->n_scans; ++t) {
    const suma_scan_ref& sc = job->scans[t];
    if (failed == SUMA_OK) {
      r = job->on_device ? suma_pipeline_begin_scan_device(s, sc.points, sc.labels, sc.probs, sc.n)
                         : suma_pipeline_begin_scan(s, sc.points, sc.labels, sc.probs, sc.n);
      if (r != SUMA_OK) fail_local(r);
    }
    if (failed != SUMA_OK && job->world == 1) break;
    int32_t win = -1;
    if (t > 0) {
      std::fill(local.begin(), local.end(), 0.0);
      if (failed == SUMA_OK && !mine.empty()) {
        for (size_t j = 0; j < mine.size(); ++j) mul4(increment, job->perturbations + 16 * (size_t)mine[j], &starts[16 * j]);
        r = suma_pipeline_minimize_hypotheses(s, starts.data(), (uint32_t)mine.size(), fixed_iterations, Ts.data(), stats.data());
        if (r != SUMA_OK) fail_local(r);
        if (failed == SUMA_OK)
          for (size_t j = 0; j < mine.size(); ++j) {
            double* row = &local[18 * (size_t)mine[j]];
            memcpy(row, &Ts[16 * j], 16 * sizeof(double));
            row[16] = stats[j].error;
            row[17] = (double)stats[j].valid;
          }
      }
      if (failed != SUMA_OK) {
        if (job->world == 1) break;
        for (uint32_t k = 0; k < n_hyp; ++k) local[18 * (size_t)k + 16] = __builtin_nan("");
      }
      if (job->world > 1) {
        r = exchange(user, local.data(), all.data(), 18 * n_hyp);
        if (r != SUMA_OK) {
          set_error(error, 160, "exchange callback failed");
          suma_pipeline_destroy(s);
          return r;
        }
        bool poisoned = false;
        for (uint32_t k = 0; k < n_hyp; ++k) poisoned = poisoned || (all[18 * (size_t)k + 16] != all[18 * (size_t)k + 16]);
        if (poisoned) { /* some rank failed: everybody leaves here, the failing rank with its own error text */
          if (failed == SUMA_OK) set_error(error, 160, "another rank reported an error in the hypothesis exchange");
          suma_pipeline_destroy(s);
          return failed != SUMA_OK ? failed : SUMA_ERR_HIP;
        }
      } else {
        all = local;
      }
      /* smallest residual per valid pair; ties -> lowest hypothesis index: the same decision on every rank */
      double best = 0.0;
      for (uint32_t k = 0; k < n_hyp; ++k) {
        const double valid = all[18 * (size_t)k + 17];
        const double v = valid <= 0.0 ? __builtin_inf() : all[18 * (size_t)k + 16] / valid;
        if (win < 0 || v < best) {
          win = (int32_t)k;
          best = v;
        }
      }
      memcpy(increment, &all[18 * (size_t)win], sizeof(increment));
      r = suma_pipeline_apply_increment(s, increment);
      if (r != SUMA_OK) fail_local(r);
    } else if (failed == SUMA_OK) {
      r = suma_pipeline_update_pose(s, fixed_iterations); /* first scan: nothing to register against (:190) */
      if (r != SUMA_OK) fail_local(r);
    }
    if (failed == SUMA_OK) {
      r = suma_pipeline_update_map(s);
      if (r != SUMA_OK) fail_local(r);
    }
    if (failed == SUMA_OK) {
      winners[t] = win;
      suma_pipeline_pose(s, poses + 16 * (size_t)t);
    }
  }
  r = failed;
  if (r == SUMA_OK) {
    r = suma_synchronize(suma_pipeline_ctx(s));
    if (r != SUMA_OK) set_error(error, 160, suma_last_error(suma_pipeline_ctx(s)));
  }
  suma_pipeline_destroy(s);
  return r;
}
