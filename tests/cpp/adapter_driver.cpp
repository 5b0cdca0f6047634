/*
 * tests/cpp/adapter_driver.cpp -- drives include/suma_adapter.hpp (the reference-side C++ binding of
 * INTEGRATION.md) the way SurfelMapping drives the reference classes, on scans read from KITTI-style .bin files:
 *   Preprocessing::process -> SurfelMap::render -> Frame2Model::setData + LieGaussNewton::minimize ->
 *   (a second objective with the fallback gates, frame to frame, as recovery_ in SurfelMapping.cpp:438-449) ->
 *   SurfelMap::update -> SurfelMap::render
 * and prints, per scan, the bit patterns of the pose and a few counters.  tests/test_gpu_cpp.py runs the same
 * sequence through the ctypes mirror and compares the lines.
 */
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "suma_adapter.hpp"

static void mul4(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      C[4 * c + r] = ((A[r] * B[4 * c] + A[4 + r] * B[4 * c + 1]) + A[8 + r] * B[4 * c + 2]) + A[12 + r] * B[4 * c + 3];
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const char* dir = argv[1];
  const int n_scans = std::atoi(argv[2]);
  const uint32_t width = (uint32_t)std::atoi(argv[3]);
  suma_params p;
  suma_params_default(&p);
  p.data_width = p.model_width = width;
  p.max_iterations = 10;
  p.label_offset = p.prob_offset = 0;
  try {
    suma_hip::Context ctx(p, 0);
    suma_hip::Preprocessing pre(ctx);
    suma_hip::SurfelMap map(ctx);
    suma_params pf = p; /* fallback_params, SurfelMapping.cpp:87-94 */
    pf.icp_max_distance = p.fallback_max_distance;
    pf.icp_max_angle = p.fallback_max_angle;
    suma_hip::Frame2Model objective(ctx), recovery(ctx, pf);
    suma_hip::LieGaussNewton gn(ctx);
    auto current = std::make_shared<suma_hip::Frame>(ctx, width, 64);
    auto last = std::make_shared<suma_hip::Frame>(ctx, width, 64);
    suma_hip::Frame model(ctx, width, 64);
    double pose[16], increment[16], increment_before[16];
    for (int i = 0; i < 16; ++i) pose[i] = increment[i] = increment_before[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < n_scans; ++k) {
      char path[4096];
      std::snprintf(path, sizeof(path), "%s/%06d.bin", dir, k);
      FILE* f = std::fopen(path, "rb");
      if (!f) return 3;
      std::fseek(f, 0, SEEK_END);
      const size_t n = (size_t)std::ftell(f) / sizeof(suma_float4);
      std::fseek(f, 0, SEEK_SET);
      std::vector<suma_float4> pts(n);
      if (std::fread(pts.data(), sizeof(suma_float4), n, f) != n) return 4;
      std::fclose(f);
      std::swap(current, last);
      pre.process(pts.data(), (uint32_t)n, *current, nullptr, nullptr, (uint32_t)k);
      float posef[16];
      for (int i = 0; i < 16; ++i) posef[i] = (float)pose[i];
      map.render(posef, model, -2.0f); /* below the log-odds of fresh surfels, like the ramp of SurfelMapping.cpp:333-340 */
      uint32_t fb_outlier = 0;
      if (k > 0) {
        objective.setData(current, map.newMapFrame());
        gn.minimize(objective, increment);
        double inc[16];
        std::memcpy(inc, gn.pose(), sizeof(inc));
        /* the fallback objective on another frame pair, with its own gates: must not disturb `objective` */
        recovery.setData(current, last);
        gn.minimize(recovery, increment);
        fb_outlier = recovery.outlier();
        double JtJ[36], Jtf[6];
        objective.initialize(inc);
        objective.jacobianProducts(JtJ, Jtf); /* re-binds objective's own frames and gates */
        std::memcpy(increment, inc, sizeof(inc));
        double np_[16];
        mul4(pose, increment, np_);
        std::memcpy(pose, np_, sizeof(np_));
      }
      for (int i = 0; i < 16; ++i) posef[i] = (float)pose[i];
      map.update(posef, *current);
      std::printf("%d", k);
      for (int i = 0; i < 16; ++i) {
        uint64_t b;
        std::memcpy(&b, &pose[i], 8);
        std::printf(" %016" PRIx64, b);
      }
      std::printf(" %u %u %u %u %u %s\n", map.size(), objective.valid(), objective.outlier(), fb_outlier,
                  map.getDataSurfels().n, gn.reason(0).c_str());
      if (k > 0) { /* LieGaussNewton::history(): fetched lazily; entry 0 is the start pose, the last one the result of `recovery` */
        const std::vector<double>& h = gn.history();
        if (h.size() < 32 || h.size() % 16 != 0) return 5;
        for (int i = 0; i < 16; ++i)
          if (h[i] != increment_before[i] || h[h.size() - 16 + i] != gn.pose()[i]) return 6;
      }
      std::memcpy(increment_before, increment, sizeof(increment));
    }
    /* ---- the history belongs to the optimizer object that minimised last on the context (LieGaussNewton.h:72 keeps it
     *      per object; here one device buffer per context, guarded by suma_icp_history_sequence) ---- */
    {
      suma_hip::LieGaussNewton gn2(ctx);
      gn.minimize(objective, increment);
      gn2.minimize(objective, increment);
      bool threw = false;
      try {
        gn.history();
      } catch (const std::runtime_error&) {
        threw = true;
      }
      if (!threw) return 13;
      if (gn2.history().size() < 32) return 14;
    }
    /* ---- SurfelMapping::Stats (SurfelMapping.cpp:183-207, 393-394): the keys the untouched GUI plots ---- */
    suma_hip::SurfelMapping sm(p, 0);
    auto nop = [](suma_hip::SurfelMapping&) {};
    for (int k = 0; k < n_scans; ++k) {
      char path[4096];
      std::snprintf(path, sizeof(path), "%s/%06d.bin", dir, k);
      FILE* f = std::fopen(path, "rb");
      if (!f) return 3;
      std::fseek(f, 0, SEEK_END);
      const size_t n = (size_t)std::ftell(f) / sizeof(suma_float4);
      std::fseek(f, 0, SEEK_SET);
      std::vector<suma_float4> pts(n);
      if (std::fread(pts.data(), sizeof(suma_float4), n, f) != n) return 4;
      std::fclose(f);
      if (k & 1)
        sm.processScan(pts.data(), nullptr, nullptr, (uint32_t)n, nop, nop, 0);
      else
        sm.processScan(pts.data(), nullptr, nullptr, (uint32_t)n, 0); /* the one-argument form fills them too */
      const std::map<std::string, double>& st = sm.getStatistics();
      const char* always[] = {"initialize-time", "preprocessing-time", "mapping-time", "map-update", "complete-time", "icp_percentage"};
      const char* tracked[] = {"icp-time", "opt-time", "icp-overall", "loop-time", "num_iterations"};
      for (const char* key : always)
        if (!st.count(key)) { std::fprintf(stderr, "missing statistics key %s\n", key); return 7; }
      if (k > 0) {
        for (const char* key : tracked)
          if (!st.count(key)) { std::fprintf(stderr, "missing statistics key %s\n", key); return 7; }
        const double it = st.at("num_iterations"), icp = st.at("icp-time"), all = st.at("complete-time");
        if (!(it >= 1.0 && it <= (double)p.max_iterations)) return 8;
        if (!(icp > 0.0 && icp <= all && all < 5.0)) return 9;
        if (!(st.at("icp_percentage") > 0.0 && st.at("icp_percentage") <= 1.0)) return 10;
        if (!(st.at("mapping-time") > 0.0 && st.at("preprocessing-time") > 0.0)) return 11;
      } else if (st.count("icp-time")) {
        return 12; /* updatePose does not run on the first scan (SurfelMapping.cpp:190) */
      }
    }
    std::printf("statistics ok: icp-time %.6f s, complete-time %.6f s, num_iterations %.0f\n", sm.getStatistics().at("icp-time"),
                sm.getStatistics().at("complete-time"), sm.getStatistics().at("num_iterations"));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
