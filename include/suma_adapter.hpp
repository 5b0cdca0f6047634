/*
 * suma_adapter.hpp -- C++ adapter that keeps the reference's class interfaces (src/core) and forwards to
 * the C-ABI of include/suma_hip.h.  A maintainer of PRBonn/semantic_suma compiles this in place of
 * the method bodies of Preprocessing.cpp / Frame2Model.cpp / LieGaussNewton.cpp / SurfelMap.cpp; the
 * caller (SurfelMapping.cpp) and everything above it stay untouched.  Header-only, C++11, no Eigen /
 * glow dependency: matrices cross as column-major pointers (Eigen::Matrix4f::data() is exactly that).
 *
 * Error behaviour mirrors the reference: failures throw std::runtime_error (Frame2Model.cpp:132,
 * Objective.h:31, SurfelMapping.cpp:98).
 */
#ifndef SUMA_ADAPTER_HPP_
#define SUMA_ADAPTER_HPP_

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "suma_hip.h"

namespace suma_hip {

inline void check(suma_ctx* c, int rc, const char* what) {
  if (rc != SUMA_OK) throw std::runtime_error(std::string(what) + ": " + suma_last_error(c));
}

/* one per GPU; shared by the adapter objects below (the reference shares one GL context the same way) */
class Context {
 public:
  explicit Context(const suma_params& p, int device = 0) : params_(p) {
    if (suma_ctx_create(&p, device, &c_) != SUMA_OK)
      throw std::runtime_error(std::string("suma_ctx_create: ") + suma_last_error(nullptr));
  }
  ~Context() { suma_ctx_destroy(c_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  suma_ctx* get() const { return c_; }
  const suma_params& params() const { return params_; }
  void setParameters(const suma_params& p) {  /* every class' setParameters(const rv::ParameterList&) */
    check(c_, suma_set_params(c_, &p), "suma_set_params");
    params_ = p;
  }

 private:
  suma_ctx* c_{nullptr};
  suma_params params_;
};

/* src/core/Frame.h:21-79 */
class Frame {
 public:
  Frame(Context& ctx, uint32_t w, uint32_t h) : ctx_(ctx), owned_(true) {
    check(ctx.get(), suma_frame_create(ctx.get(), w, h, &f_), "suma_frame_create");
  }
  Frame(Context& ctx, suma_frame* borrowed) : ctx_(ctx), f_(borrowed), owned_(false) {}
  ~Frame() {
    if (owned_) suma_frame_destroy(f_);
  }
  void copy(const Frame& other) { check(ctx_.get(), suma_frame_copy(ctx_.get(), f_, other.f_), "Frame::copy"); }
  void download(int which, std::vector<suma_float4>& out) const {
    out.resize((size_t)suma_frame_width(f_) * suma_frame_height(f_));
    check(ctx_.get(), suma_frame_download(ctx_.get(), f_, which, out.data()), "suma_frame_download");
  }
  suma_frame* get() const { return f_; }
  float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; /* Frame::pose, Frame.h:74 */
  bool valid{false};

 private:
  Context& ctx_;
  suma_frame* f_{nullptr};
  bool owned_;
};

/* src/core/Preprocessing.h:47-58 */
class Preprocessing {
 public:
  explicit Preprocessing(Context& ctx) : ctx_(ctx) {}
  /* reference: process(GlBuffer<rv::Point3f>& points, Frame&, GlBuffer<float>& labels, GlBuffer<float>& probs, t) */
  void process(const suma_float4* points, uint32_t n, Frame& frame, const float* labels, const float* probs,
               uint32_t timestamp) {
    check(ctx_.get(), suma_preprocess(ctx_.get(), points, labels, probs, n, timestamp, frame.get()),
          "Preprocessing::process");
    frame.valid = true;
  }

 private:
  Context& ctx_;
};

/* src/core/Objective.h:14-82 as implemented by src/core/Frame2Model.h:28-73 */
class Frame2Model {
 public:
  explicit Frame2Model(Context& ctx) : ctx_(ctx) { std::memset(&stats_, 0, sizeof(stats_)); eye(pose_); }
  uint32_t num_parameters() const { return 6; }
  void setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last) {
    current_ = current;
    last_ = last;
    check(ctx_.get(), suma_icp_set_data(ctx_.get(), current->get(), last->get()), "Frame2Model::setData");
    iteration_ = 0;
  }
  void initialize(const double* pose16) { std::memcpy(pose_, pose16, sizeof(pose_)); iteration_ = 0; }
  /* returns F; JtJ 6x6 column-major, Jtf 6 (Eigen::MatrixXd::data() of the reference's arguments) */
  double jacobianProducts(double* JtJ, double* Jtf) {
    check(ctx_.get(), suma_icp_jacobian_products(ctx_.get(), pose_, iteration_, JtJ, Jtf, nullptr, &stats_),
          "Frame2Model::jacobianProducts");
    iteration_ += 1;
    return stats_.error;
  }
  const double* pose() const { return pose_; }
  uint32_t inlier() const { return stats_.inlier; }
  uint32_t outlier() const { return stats_.outlier; }
  uint32_t valid() const { return stats_.valid; }
  uint32_t invalid() const { return stats_.invalid; }
  double inlier_residual() const { return stats_.inlier_residual; }
  uint32_t getMaxLevel() const { return 0; } /* Frame2Model.cpp:127-129 */

 private:
  friend class LieGaussNewton;
  static void eye(double* T) { for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  Context& ctx_;
  std::shared_ptr<Frame> current_, last_;
  double pose_[16];
  uint32_t iteration_{0};
  suma_icp_stats stats_;
};

/* src/core/LieGaussNewton.h:25-76: the whole loop runs on the device */
class LieGaussNewton {
 public:
  explicit LieGaussNewton(Context& ctx) : ctx_(ctx) { std::memset(&stats_, 0, sizeof(stats_)); }
  int32_t minimize(Frame2Model& F, const double* T0) {
    history_.assign(16 * 1025, 0.0);
    uint32_t nh = 0;
    check(ctx_.get(), suma_icp_minimize(ctx_.get(), T0, pose_, history_.data(), 1025, &nh, &stats_),
          "LieGaussNewton::minimize");
    history_.resize(16 * (size_t)(nh < 1025 ? nh : 1025));
    std::memcpy(F.pose_, pose_, sizeof(pose_));
    F.stats_ = stats_;
    return 0;
  }
  const double* pose() const { return pose_; }
  const std::vector<double>& history() const { return history_; } /* 16 doubles per entry */
  uint32_t iterationCount() const { return stats_.iterations; }

 private:
  Context& ctx_;
  double pose_[16];
  std::vector<double> history_;
  suma_icp_stats stats_;
};

/* src/core/SurfelMap.h:36-78 (draw / setColorMap are visualisation and stay with the GL code) */
class SurfelMap {
 public:
  explicit SurfelMap(Context& ctx) : ctx_(ctx) {}
  void reset() { check(ctx_.get(), suma_map_reset(ctx_.get()), "SurfelMap::reset"); }
  void update(const float* pose, Frame& frame) {
    check(ctx_.get(), suma_map_update(ctx_.get(), pose, frame.get()), "SurfelMap::update");
  }
  void render(const float* pose, Frame& frame, float ct) { render(pose, pose, frame, ct); }
  void render(const float* pose_old, const float* pose_new, Frame& frame, float ct) {
    check(ctx_.get(), suma_map_render(ctx_.get(), pose_old, pose_new, ct, frame.get()), "SurfelMap::render");
  }
  void render_active(const float* pose, float ct) {
    check(ctx_.get(), suma_map_render_active(ctx_.get(), pose, ct), "SurfelMap::render_active");
  }
  void render_inactive(const float* pose, float ct) {
    check(ctx_.get(), suma_map_render_inactive(ctx_.get(), pose, ct), "SurfelMap::render_inactive");
  }
  void render_composed(const float* pose_old, const float* pose_new, float ct) {
    check(ctx_.get(), suma_map_render_composed(ctx_.get(), pose_old, pose_new, ct), "SurfelMap::render_composed");
  }
  std::shared_ptr<Frame> oldMapFrame() { return borrowed(SUMA_FRAME_OLD); }
  std::shared_ptr<Frame> newMapFrame() { return borrowed(SUMA_FRAME_NEW); }
  std::shared_ptr<Frame> composedFrame() { return borrowed(SUMA_FRAME_COMPOSED); }
  void updatePoses(const std::vector<float>& poses16) {
    check(ctx_.get(), suma_map_update_poses(ctx_.get(), poses16.data(), (uint32_t)(poses16.size() / 16)),
          "SurfelMap::updatePoses");
  }
  uint32_t size() const {
    uint32_t n = 0;
    check(ctx_.get(), suma_map_size(ctx_.get(), &n), "SurfelMap::size");
    return n;
  }
  std::vector<suma_surfel> getAllSurfels() {
    std::vector<suma_surfel> out(size());
    uint32_t n = 0;
    check(ctx_.get(), suma_map_download(ctx_.get(), out.data(), (uint32_t)out.size(), &n), "SurfelMap::getAllSurfels");
    return out;
  }

 private:
  std::shared_ptr<Frame> borrowed(int which) {
    return std::make_shared<Frame>(ctx_, suma_map_frame(ctx_.get(), which));
  }
  Context& ctx_;
};

}  // namespace suma_hip
#endif
