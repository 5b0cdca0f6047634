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
    double pose[16], increment[16];
    for (int i = 0; i < 16; ++i) pose[i] = increment[i] = (i % 5 == 0) ? 1.0 : 0.0;
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
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
