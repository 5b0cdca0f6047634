/*
 * tools/adapter_bench.cpp -- what does a host pay that does NOT use the scan pipeline?
 *
 * Runs the same scans (read from a directory written by bench.py: NNNNNN.bin points, .lab / .prob float32 arrays)
 * three ways through include/suma_adapter.hpp, all from host vectors as the reference's caller holds them:
 *   classes   the class-by-class sequence of SurfelMapping::processScan without loop closures
 *             (SurfelMapping.cpp:175-210, 323-358, 372-476, 797-804): Preprocessing::process, SurfelMap::render,
 *             Frame2Model::setData + LieGaussNewton::minimize, render_active + Frame::copy + jacobianProducts
 *             (the statistics pass), the fallback decision, SurfelMap::update, SurfelMap::render -- every call
 *             synchronous, as the reference's GL calls are;
 *   phases    suma_hip::SurfelMapping::processScan with (empty) loop-closure hooks = begin_scan / update_pose /
 *             update_map: what a host with close-loops = true runs;
 *   pipeline  suma_pipeline_process_scan in one call.
 * Prints one JSON object.  usage: adapter_bench <dir> <n_scans> <width> <height> <gn_iterations>
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "suma_adapter.hpp"

struct Scan {
  std::vector<suma_float4> pts;
  std::vector<float> lab, prob;
};

static bool read_file(const char* path, void* dst, size_t bytes) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  const bool ok = std::fread(dst, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok;
}
static size_t file_size(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return 0;
  std::fseek(f, 0, SEEK_END);
  const size_t n = (size_t)std::ftell(f);
  std::fclose(f);
  return n;
}
static void mul4(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] = ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}
static void rigid_inv(const double* m, double* out) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) out[4 * c + r] = m[4 * r + c];
  for (int r = 0; r < 3; ++r) out[12 + r] = -((m[4 * r] * m[12] + m[4 * r + 1] * m[13]) + m[4 * r + 2] * m[14]);
  out[3] = out[7] = out[11] = 0.0;
  out[15] = 1.0;
}
static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* SurfelMapping::getConfidenceThreshold, SurfelMapping.cpp:333-340 */
static float conf_threshold(const suma_params& p, uint32_t t) {
  float ct = p.confidence_threshold;
  if (t < 10) {
    const float pu = 0.1f, log_unstable = (float)std::log((double)(pu / (1.0f - pu)));
    const float alpha = (float)t / 10.0f;
    ct = (float)((1.0 - (double)alpha) * (double)log_unstable + (double)(alpha * p.confidence_threshold));
  }
  return ct;
}

/* the reference's processScan on the adapter classes, one synchronous call after the other */
static double run_classes(const suma_params& p0, const std::vector<Scan>& scans, size_t preroll, int gn_iterations,
                          double* end_pose) {
  suma_params p = p0;
  p.max_iterations = (uint32_t)gn_iterations;
  p.stopping_threshold = 0.0f;
  p.delta = 0.0f;
  suma_hip::Context ctx(p, 0);
  suma_hip::Preprocessing pre(ctx);
  suma_hip::SurfelMap map(ctx);
  suma_params pf = p;
  pf.icp_max_distance = p.fallback_max_distance;
  pf.icp_max_angle = p.fallback_max_angle;
  suma_hip::Frame2Model objective(ctx), recovery(ctx, pf);
  suma_hip::LieGaussNewton gn(ctx);
  auto current = std::make_shared<suma_hip::Frame>(ctx, p.data_width, p.data_height);
  auto last = std::make_shared<suma_hip::Frame>(ctx, p.data_width, p.data_height);
  auto current_model = std::make_shared<suma_hip::Frame>(ctx, p.model_width, p.model_height);
  auto last_model = std::make_shared<suma_hip::Frame>(ctx, p.model_width, p.model_height);
  double pose[16], increment[16], I[16];
  for (int i = 0; i < 16; ++i) pose[i] = increment[i] = I[i] = (i % 5 == 0) ? 1.0 : 0.0;
  float posef[16];
  double t0 = now();
  for (uint32_t k = 0; k < scans.size(); ++k) {
    if (k == preroll) { /* the timed stretch starts on a drained context, like the pipeline modes below */
      suma_synchronize(ctx.get());
      t0 = now();
    }
    const Scan& sc = scans[k];
    std::swap(current, last); /* initialize(), :323-331 */
    std::swap(current_model, last_model);
    pre.process(sc.pts.data(), (uint32_t)sc.pts.size(), *current, sc.lab.data(), sc.prob.data(), k); /* :344 */
    for (int i = 0; i < 16; ++i) posef[i] = (float)pose[i];
    map.render(posef, posef, *last_model, conf_threshold(p, k)); /* :351 */
    if (k > 0) {
      objective.setData(current, map.newMapFrame()); /* :384 */
      gn.minimize(objective, increment);            /* :390 */
      double inc[16], delta[16], inv_last[16], posed[16];
      std::memcpy(inc, gn.pose(), sizeof(inc));
      rigid_inv(increment, inv_last);
      mul4(inv_last, inc, delta);
      mul4(pose, inc, posed);
      float pf32[16];
      for (int i = 0; i < 16; ++i) pf32[i] = (float)posed[i];
      map.render_active(pf32, conf_threshold(p, k));    /* :406 */
      last_model->copy(*map.newMapFrame());            /* :407 */
      objective.setData(current, map.newMapFrame());   /* :408 */
      objective.initialize(I);                         /* :411 */
      double JtJ[36], Jtr[6];
      objective.jacobianProducts(JtJ, Jtr);            /* :413 */
      const float t_err = (float)std::sqrt((delta[12] * delta[12] + delta[13] * delta[13]) + delta[14] * delta[14]);
      const float angle = (float)(0.5 * (((delta[0] + delta[5]) + delta[10]) - 1.0));
      const float r_err = (float)std::acos((double)std::fmax(std::fmin(angle, 1.0f), -1.0f));
      if (k > 1 && ((double)t_err > 0.4 || (double)r_err > 0.1) && p.fallback_mode) { /* :438-449 */
        recovery.setData(current, last);
        gn.minimize(recovery, increment);
        std::memcpy(inc, gn.pose(), sizeof(inc));
      }
      double np_[16];
      mul4(pose, inc, np_);
      std::memcpy(pose, np_, sizeof(np_));
      std::memcpy(increment, inc, sizeof(inc));
    }
    for (int i = 0; i < 16; ++i) posef[i] = (float)pose[i];
    map.update(posef, *current);                                   /* :799 */
    map.render(posef, posef, *current_model, conf_threshold(p, k)); /* :803; the reference's threshold is read after ... */
  }
  suma_synchronize(ctx.get());
  const double dt = now() - t0;
  std::memcpy(end_pose, pose, sizeof(pose));
  return dt;
}

/* mode 0: one call per scan from host vectors; 1: the phase calls with empty loop-closure hooks; 2: scans resident
 * in HBM beforehand (suma_pipeline_process_scan_device) -- the rate the host-vector entries are measured against */
static double run_pipeline(const suma_params& p, const std::vector<Scan>& scans, size_t preroll, int gn_iterations, int mode,
                           double* end_pose) {
  suma_hip::SurfelMapping sm(p, 0);
  auto nop = [](suma_hip::SurfelMapping&) {};
  const bool phases = mode == 1;
  std::vector<void*> dev;
  if (mode == 2) {
    for (const Scan& sc : scans) {
      void *dp = nullptr, *dl = nullptr, *dq = nullptr;
      const size_t n = sc.pts.size();
      suma_hip::check(sm.ctx(), suma_device_alloc(sm.ctx(), n * sizeof(suma_float4), &dp), "alloc");
      suma_hip::check(sm.ctx(), suma_device_alloc(sm.ctx(), n * sizeof(float), &dl), "alloc");
      suma_hip::check(sm.ctx(), suma_device_alloc(sm.ctx(), n * sizeof(float), &dq), "alloc");
      suma_device_upload(sm.ctx(), dp, sc.pts.data(), n * sizeof(suma_float4));
      suma_device_upload(sm.ctx(), dl, sc.lab.data(), n * sizeof(float));
      suma_device_upload(sm.ctx(), dq, sc.prob.data(), n * sizeof(float));
      dev.push_back(dp);
      dev.push_back(dl);
      dev.push_back(dq);
    }
    suma_synchronize(sm.ctx());
    double t0 = now();
    for (size_t k = 0; k < scans.size(); ++k) {
      if (k == preroll) {
        suma_synchronize(sm.ctx());
        t0 = now();
      }
      suma_hip::check(sm.ctx(), suma_pipeline_process_scan_device(sm.get(), (const suma_float4*)dev[3 * k], (const float*)dev[3 * k + 1],
                                                                   (const float*)dev[3 * k + 2], (uint32_t)scans[k].pts.size(),
                                                                   gn_iterations), "process_scan_device");
    }
    suma_synchronize(sm.ctx());
    const double dt = now() - t0;
    sm.getCurrentPose(end_pose);
    for (void* d : dev) suma_device_free(sm.ctx(), d);
    return dt;
  }
  double t0 = now();
  for (size_t k = 0; k < scans.size(); ++k) {
    const Scan& sc = scans[k];
    if (k == preroll) {
      suma_synchronize(sm.ctx());
      t0 = now();
    }
    if (phases)
      sm.processScan(sc.pts.data(), sc.lab.data(), sc.prob.data(), (uint32_t)sc.pts.size(), nop, nop, gn_iterations);
    else
      suma_hip::check(sm.ctx(), suma_pipeline_process_scan(sm.get(), sc.pts.data(), sc.lab.data(), sc.prob.data(),
                                                             (uint32_t)sc.pts.size(), gn_iterations), "suma_pipeline_process_scan");
  }
  suma_synchronize(sm.ctx());
  const double dt = now() - t0;
  sm.getCurrentPose(end_pose);
  return dt;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const char* dir = argv[1];
  const int n_scans = std::atoi(argv[2]);
  suma_params p;
  suma_params_default(&p);
  if (std::getenv("SUMA_ADAPTER_MAX_SURFELS")) p.max_surfels = (uint32_t)std::atol(std::getenv("SUMA_ADAPTER_MAX_SURFELS"));
  p.data_width = p.model_width = (uint32_t)std::atoi(argv[3]);
  p.data_height = p.model_height = (uint32_t)std::atoi(argv[4]);
  const int gn_iterations = std::atoi(argv[5]);
  const size_t preroll = (argc > 6 && std::atoi(argv[6]) > 0 && std::atoi(argv[6]) < n_scans) ? (size_t)std::atoi(argv[6]) : 0;
  std::vector<Scan> scans((size_t)n_scans);
  for (int k = 0; k < n_scans; ++k) {
    char path[4096];
    std::snprintf(path, sizeof(path), "%s/%06d.bin", dir, k);
    const size_t n = file_size(path) / sizeof(suma_float4);
    if (!n) return 3;
    scans[k].pts.resize(n);
    scans[k].lab.resize(n);
    scans[k].prob.resize(n);
    if (!read_file(path, scans[k].pts.data(), n * sizeof(suma_float4))) return 3;
    std::snprintf(path, sizeof(path), "%s/%06d.lab", dir, k);
    if (!read_file(path, scans[k].lab.data(), n * sizeof(float))) return 3;
    std::snprintf(path, sizeof(path), "%s/%06d.prob", dir, k);
    if (!read_file(path, scans[k].prob.data(), n * sizeof(float))) return 3;
  }
  try {
    double pc[16], pp[16], pl[16];
    /* a short run of each first (module load, first-touch allocations), then the timed runs */
    std::vector<Scan> head(scans.begin(), scans.begin() + (n_scans < 5 ? n_scans : 5));
    run_classes(p, head, 0, gn_iterations, pc);
    run_pipeline(p, head, 0, gn_iterations, 1, pp);
    double pr[16];
    const double t_classes = run_classes(p, scans, preroll, gn_iterations, pc);
    const double t_phases = run_pipeline(p, scans, preroll, gn_iterations, 1, pp);
    const double t_pipeline = run_pipeline(p, scans, preroll, gn_iterations, 0, pl);
    const double t_resident = run_pipeline(p, scans, preroll, gn_iterations, 2, pr);
    const int n_timed = n_scans - (int)preroll;
    bool same = true;
    for (int i = 0; i < 16; ++i) same = same && pc[i] == pp[i] && pp[i] == pl[i] && pl[i] == pr[i];
    std::printf("{\"scans\": %d, \"after_scans\": %d, \"classes_scans_per_s\": %.1f, \"phases_scans_per_s\": %.1f, "
                "\"pipeline_scans_per_s\": %.1f, \"resident_scans_per_s\": %.1f, \"classes_vs_phases\": %.3f, "
                "\"host_vectors_vs_resident\": %.3f, \"end_pose_bits_equal\": %s, \"input\": \"host vectors (pageable) unless "
                "resident, %ux%u, %d GN iterations, scans %d..%d of the sequence (each mode replays scans 0..%d untimed "
                "first)\"}\n",
                n_timed, (int)preroll, n_timed / t_classes, n_timed / t_phases, n_timed / t_pipeline, n_timed / t_resident,
                t_phases / t_classes, t_resident / t_pipeline, same ? "true" : "false", p.data_width, p.data_height,
                gn_iterations, (int)preroll, n_scans - 1, (int)preroll - 1);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_bench: %s\n", e.what());
    return 1;
  }
  return 0;
}
