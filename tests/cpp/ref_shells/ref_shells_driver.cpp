/*
 * tests/cpp/ref_shells/ref_shells_driver.cpp -- SurfelMapping's per-scan call sequence (SurfelMapping.cpp:175-210,
 * 323-358, 372-476, 797-804; loop closures off) written against the reference's class interfaces exactly as
 * SurfelMapping.cpp writes it -- rv::ParameterList in the constructors, glow::GlBuffer scan buffers, Eigen matrices,
 * std::shared_ptr<Frame> swaps -- on the shells of core_shells.h / ref_shells.cpp.
 *
 *   ref_shells_driver --params          CPU: the rv::ParameterList of config/default.xml -> suma_params, printed
 *   ref_shells_driver <dir> <n> <W>     GPU: n scans from KITTI-style .bin files; prints the pose bits per scan
 */
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "core_shells.h"

/* config/default.xml as the rv::ParameterList that parseXmlFile builds from it (the keys of SURVEY.md Appendix C) */
static rv::ParameterList default_xml(uint32_t width) {
  rv::ParameterList p;
  p.insert(rv::IntegerParameter("data_width", (int)width));
  p.insert(rv::IntegerParameter("data_height", 64));
  p.insert(rv::FloatParameter("data_fov_up", 3.0));
  p.insert(rv::FloatParameter("data_fov_down", -25.0));
  p.insert(rv::FloatParameter("max_depth", 75.0));
  p.insert(rv::FloatParameter("min_depth", 2.0));
  p.insert(rv::IntegerParameter("max iterations", 33));
  p.insert(rv::FloatParameter("stopping threshold", 0.0001));
  p.insert(rv::FloatParameter("delta", 0.0001));
  p.insert(rv::FloatParameter("icp-max-distance", 2.0));
  p.insert(rv::FloatParameter("icp-max-angle", 30.0));
  p.insert(rv::StringParameter("weighting", "huber"));
  p.insert(rv::FloatParameter("factor", 0.5));
  p.insert(rv::BooleanParameter("initialize_identity", false));
  p.insert(rv::BooleanParameter("bilinear_sampling", true));
  p.insert(rv::FloatParameter("cutoff_threshold", 10.0));
  p.insert(rv::IntegerParameter("model_width", (int)width));
  p.insert(rv::IntegerParameter("model_height", 64));
  p.insert(rv::FloatParameter("model_fov_up", 3.0));
  p.insert(rv::FloatParameter("model_fov_down", -25.0));
  p.insert(rv::FloatParameter("model_max_depth", 75.0));
  p.insert(rv::FloatParameter("model_min_depth", 2.0));
  p.insert(rv::BooleanParameter("compose_rendering", true));
  p.insert(rv::FloatParameter("max_loop_closure_distance", 8.0));
  p.insert(rv::BooleanParameter("fallback_mode", true));
  p.insert(rv::FloatParameter("fallback-max-distance", 0.5));
  p.insert(rv::FloatParameter("fallback-max-angle", 30.0));
  p.insert(rv::StringParameter("approach", "frame-to-model"));
  p.insert(rv::FloatParameter("min_radius", 0.03));
  p.insert(rv::FloatParameter("max_radius", 1.00));
  p.insert(rv::FloatParameter("max_angle", 90.0));
  p.insert(rv::FloatParameter("map-max-distance", 0.2));
  p.insert(rv::FloatParameter("map-max-angle", 45.0));
  p.insert(rv::IntegerParameter("unstable_age", 3));
  p.insert(rv::IntegerParameter("confidence_mode", 3));
  p.insert(rv::FloatParameter("confidence_threshold", 0.0));
  p.insert(rv::FloatParameter("p_stable", 0.6));
  p.insert(rv::FloatParameter("p_prior", 0.5));
  p.insert(rv::FloatParameter("sigma_angle", 1.0));
  p.insert(rv::FloatParameter("sigma_distance", 1.0));
  p.insert(rv::BooleanParameter("use_stability", true));
  p.insert(rv::IntegerParameter("submap-dimension", 4));
  p.insert(rv::FloatParameter("submap-extent", 10.0));
  p.insert(rv::BooleanParameter("partial-extraction", true));
  p.insert(rv::BooleanParameter("close-loops", true));
  p.insert(rv::IntegerParameter("averaging_scheme", 0));
  p.insert(rv::FloatParameter("bilateral_sigma_range", 2.5));
  p.insert(rv::BooleanParameter("update_always", false));
  p.insert(rv::BooleanParameter("use_filtered_vertexmap", false));
  p.insert(rv::IntegerParameter("weighting_scheme", 0));
  return p;
}

static int print_params() {
  const suma_params p = suma_shell::params_from(default_xml(900));
  suma_params d;
  suma_params_default(&d); /* include/suma_types.h: the same file by hand */
  d.data_width = d.model_width = 900;
  /* every field of the POD, as text: the test compares the two lines */
  const suma_params* both[2] = {&p, &d};
  for (int k = 0; k < 2; ++k) {
    const suma_params& q = *both[k];
    std::printf("%u %u %g %g %g %g %u %u %g %g %g %g %u %g %g %g %g %d %g %d %d %d %g %g %d %g %g %g %g %g %g %d %d %g %g %g %g "
                "%g %d %d %g %d %d %d %d %g %d %d %d %d\n",
                q.data_width, q.data_height, q.data_fov_up, q.data_fov_down, q.min_depth, q.max_depth, q.model_width,
                q.model_height, q.model_fov_up, q.model_fov_down, q.model_min_depth, q.model_max_depth, q.max_iterations,
                q.stopping_threshold, q.delta, q.icp_max_distance, q.icp_max_angle, q.weight_function, q.factor,
                q.bilinear_sampling, q.initialize_identity, q.fallback_mode, q.fallback_max_distance, q.fallback_max_angle,
                q.compose_rendering, q.max_loop_closure_distance, q.min_radius, q.max_radius, q.max_angle, q.map_max_distance,
                q.map_max_angle, q.unstable_age, q.confidence_mode, q.confidence_threshold, q.p_stable, q.p_prior,
                q.sigma_angle, q.sigma_distance, q.use_stability, q.active_timestamps, q.max_weight, q.weighting_scheme,
                q.averaging_scheme, q.update_always, q.submap_dimension, q.submap_extent, q.partial_extraction,
                q.avg_vertexmap, q.filter_vertexmap, q.use_filtered_vertexmap);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::strcmp(argv[1], "--params") == 0) return print_params();
  if (argc < 4) return 2;
  const char* dir = argv[1];
  const int n_scans = std::atoi(argv[2]);
  const uint32_t width = (uint32_t)std::atoi(argv[3]);
  try {
    rv::ParameterList params = default_xml(width);
    /* ---- SurfelMapping::SurfelMapping (SurfelMapping.cpp:18-70): the members of the hot path ---- */
    Preprocessing preprocessor_(params);
    std::shared_ptr<Frame> currentFrame_(new Frame(width, 64)), lastFrame_(new Frame(width, 64));
    std::shared_ptr<Frame> currentModelFrame_(new Frame(width, 64)), lastModelFrame_(new Frame(width, 64));
    std::shared_ptr<SurfelMap> map_(new SurfelMap(params));
    std::shared_ptr<Objective> objective_(new Frame2Model(params));
    std::shared_ptr<LieGaussNewton> gn_(new LieGaussNewton());
    gn_->setParameters(params);
    glow::GlBuffer<rv::Point3f> current_pts_{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};
    glow::GlBuffer<float> current_labels_{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};
    glow::GlBuffer<float> current_probs_{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};
    Eigen::Matrix4d currentPose_ = Eigen::Matrix4d::Identity(), lastIncrement_ = Eigen::Matrix4d::Identity();
    const float confidence_threshold_ = params["confidence_threshold"];
    const float log_unstable_ = std::log(0.1f / (1.0f - 0.1f)); /* SurfelMapping.cpp:108-110 */
    const uint32_t time_init = 10;
    for (uint32_t timestamp_ = 0; timestamp_ < (uint32_t)n_scans; ++timestamp_) {
      char path[4096];
      std::snprintf(path, sizeof(path), "%s/%06u.bin", dir, timestamp_);
      FILE* f = std::fopen(path, "rb");
      if (!f) return 3;
      std::fseek(f, 0, SEEK_END);
      const size_t n = (size_t)std::ftell(f) / sizeof(rv::Point3f);
      std::fseek(f, 0, SEEK_SET);
      std::vector<rv::Point3f> pts(n);
      if (std::fread(pts.data(), sizeof(rv::Point3f), n, f) != n) return 4;
      std::fclose(f);
      std::vector<float> labels_float(n, 0.0f), labels_prob(n, 0.0f);
      /* ---- initialize(scan), :323-331 ---- */
      lastFrame_.swap(currentFrame_);
      lastModelFrame_.swap(currentModelFrame_);
      current_pts_.assign(pts);
      current_labels_.assign(labels_float);
      current_probs_.assign(labels_prob);
      /* ---- preprocess(), :342-358 ---- */
      float ct = confidence_threshold_; /* getConfidenceThreshold, :333-340 */
      if (timestamp_ < time_init) {
        float alpha = float(timestamp_) / float(time_init);
        ct = (1.0 - alpha) * log_unstable_ + alpha * confidence_threshold_;
      }
      preprocessor_.process(current_pts_, *currentFrame_, current_labels_, current_probs_, timestamp_);
      map_->render(currentPose_.cast<float>(), currentPose_.cast<float>(), *lastModelFrame_, ct);
      lastModelFrame_->pose = currentPose_.cast<float>();
      /* ---- updatePose(), :372-476 ---- */
      if (timestamp_ > 0) {
        const Eigen::Matrix4d T0 = lastIncrement_;
        objective_->setData(currentFrame_, map_->newMapFrame());
        int32_t success = gn_->minimize(*objective_, T0);
        if (success < -1) std::fprintf(stderr, "%s\n", gn_->reason(success).c_str());
        const std::vector<Eigen::Matrix4d>& odom_poses_ = gn_->history();
        if (odom_poses_.empty()) return 5;
        Eigen::Matrix4d increment = gn_->pose();
        Eigen::MatrixXd JtJ_new(6, 6), Jtr(6, 1);
        map_->render_active((currentPose_ * increment).cast<float>(), ct);
        lastModelFrame_->copy(*map_->newMapFrame());
        objective_->setData(currentFrame_, map_->newMapFrame());
        objective_->initialize(Eigen::Matrix4d::Identity());
        const double residual = objective_->jacobianProducts(JtJ_new, Jtr);
        (void)residual;
        lastIncrement_ = increment;
        currentPose_ = currentPose_ * increment; /* :454 */
      }
      /* ---- updateMap(), :797-804 ---- */
      map_->update(currentPose_.cast<float>(), *currentFrame_);
      map_->render(currentPose_.cast<float>(), *currentModelFrame_, ct);
      std::printf("%u", timestamp_);
      for (int i = 0; i < 16; ++i) {
        uint64_t b;
        std::memcpy(&b, &currentPose_.data()[i], 8);
        std::printf(" %016" PRIx64, b);
      }
      std::printf(" %u %u %u %u %u\n", map_->size(), objective_->valid(), objective_->outlier(), objective_->invalid(),
                  gn_->iterationCount());
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "ref_shells_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
