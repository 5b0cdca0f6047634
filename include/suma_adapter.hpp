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

#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
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

/* SE3::exp (src/core/lie_algebra.cpp:4-34) for Objective::increment on the host side of the adapter; column-major */
inline void se3_exp(const double* x, double* T) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  const double v[3] = {x[0], x[1], x[2]}, o[3] = {x[3], x[4], x[5]};
  const double theta = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
  if (theta > 1e-10) {
    const double K[9] = {0, -o[2], o[1], o[2], 0, -o[0], -o[1], o[0], 0}; /* row-major skew matrix */
    double K2[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    const double alpha = std::sin(theta) / theta, beta = (1 - std::cos(theta)) / (theta * theta);
    const double delta = (theta - std::sin(theta)) / (theta * theta * theta);
    for (int r = 0; r < 3; ++r) {
      double t = 0.0;
      for (int c = 0; c < 3; ++c) {
        const double I = (r == c) ? 1.0 : 0.0;
        T[4 * c + r] = I + alpha * K[3 * r + c] + beta * K2[3 * r + c];
        t += (I + beta * K[3 * r + c] + delta * K2[3 * r + c]) * v[c];
      }
      T[12 + r] = t;
    }
  } else {
    T[12] = v[0];
    T[13] = v[1];
    T[14] = v[2];
  }
}

/* src/core/Objective.h:14-82 as implemented by src/core/Frame2Model.h:28-73.
 * Like the reference's object, an instance OWNS its parameters (Frame2Model::updateParameters,
 * Frame2Model.cpp:65-110) and its frame pair and sends both to the device before every launch: objective_ and
 * recovery_ = Frame2Model(fallback_params) (SurfelMapping.cpp:87-94, used at :438-449) can share one Context. */
class Frame2Model {
 public:
  explicit Frame2Model(Context& ctx) : Frame2Model(ctx, ctx.params()) {}
  /* Frame2Model(const rv::ParameterList&): the parameter block this objective is built from */
  Frame2Model(Context& ctx, const suma_params& p) : ctx_(ctx) {
    std::memset(&stats_, 0, sizeof(stats_));
    eye(pose_);
    objective_.icp_max_distance = p.icp_max_distance;
    objective_.icp_max_angle = p.icp_max_angle;
    objective_.weight_function = p.weight_function;
    objective_.factor = p.factor;
    objective_.bilinear_sampling = p.bilinear_sampling;
  }
  uint32_t num_parameters() const { return 6; }
  /* Objective::setParameter(const rv::Parameter&) -> Frame2Model::setParameter (Frame2Model.cpp:112-115): the keys
   * this objective reads; "weighting" takes huber / turkey / stability / anything else = none */
  void setParameter(const std::string& name, double value) {
    if (name == "icp-max-distance") objective_.icp_max_distance = (float)value;
    else if (name == "icp-max-angle") objective_.icp_max_angle = (float)value;
    else if (name == "factor") objective_.factor = (float)value;
    else if (name == "bilinear_sampling") objective_.bilinear_sampling = value != 0.0;
  }
  void setParameter(const std::string& name, const std::string& value) {
    if (name != "weighting") return;
    objective_.weight_function = value == "huber" ? SUMA_WEIGHT_HUBER
                               : value == "turkey" ? SUMA_WEIGHT_TUKEY
                               : value == "stability" ? SUMA_WEIGHT_STABILITY : SUMA_WEIGHT_NONE;
  }
  void setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last) {
    current_ = current;
    last_ = last;
    iteration_ = 0;
  }
  /* Objective::initialize (Objective.h:58) sets the pose and nothing else: iteration_ is reset by setData only */
  void initialize(const double* pose16) { std::memcpy(pose_, pose16, sizeof(pose_)); }
  /* Objective::residual is "not implemented" in the reference too (Frame2Model.cpp:131-134) */
  double residual(const double* /*delta6*/) { throw std::runtime_error("not implemented."); }
  /* returns F; JtJ 6x6 column-major, Jtf 6 (Eigen::MatrixXd::data() of the reference's arguments) */
  double jacobianProducts(double* JtJ, double* Jtf) {
    bind();
    check(ctx_.get(), suma_icp_jacobian_products(ctx_.get(), pose_, iteration_, JtJ, Jtf, nullptr, &stats_),
          "Frame2Model::jacobianProducts");
    return stats_.error;
  }
  /* Objective::increment (Objective.h:45-48): pose_ = SE3::exp(delta) * pose_ */
  void increment(const double* delta6) {
    double E[16], P[16];
    se3_exp(delta6, E);
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r)
        P[4 * c + r] = E[r] * pose_[4 * c] + E[4 + r] * pose_[4 * c + 1] + E[8 + r] * pose_[4 * c + 2] + E[12 + r] * pose_[4 * c + 3];
    std::memcpy(pose_, P, sizeof(P));
    iteration_ += 1;
  }
  const double* pose() const { return pose_; }
  uint32_t inlier() const { return stats_.inlier; }
  uint32_t outlier() const { return stats_.outlier; }
  uint32_t valid() const { return stats_.valid; }
  uint32_t invalid() const { return stats_.invalid; }
  double inlier_residual() const { return stats_.inlier_residual; }
  void setLevel(uint32_t) {}                 /* Frame2Model.cpp:125 */
  uint32_t getMaxLevel() const { return 0; } /* Frame2Model.cpp:127-129 */
  void reset() {}

 private:
  friend class LieGaussNewton;
  static void eye(double* T) { for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  /* this object's frames and parameters become the ones the next launch uses */
  void bind() {
    if (!current_ || !last_) throw std::runtime_error("Frame2Model::setData has not been called");
    check(ctx_.get(), suma_icp_set_data(ctx_.get(), current_->get(), last_->get()), "Frame2Model::setData");
    check(ctx_.get(), suma_icp_set_objective(ctx_.get(), &objective_), "Frame2Model parameters");
  }
  Context& ctx_;
  std::shared_ptr<Frame> current_, last_;
  suma_icp_objective objective_;
  double pose_[16];
  uint32_t iteration_{0};
  suma_icp_stats stats_;
};

/* src/core/LieGaussNewton.h:25-76: the whole loop runs on the device */
class LieGaussNewton {
 public:
  static const int32_t CONVERGED = 0;
  explicit LieGaussNewton(Context& ctx) : ctx_(ctx) { std::memset(&stats_, 0, sizeof(stats_)); }
  int32_t minimize(Frame2Model& F, const double* T0) {
    F.bind();
    /* Frame2Model::iteration_ runs on across minimisations on one setData (SurfelMapping.cpp:693-700) */
    check(ctx_.get(), suma_icp_set_iteration(ctx_.get(), F.iteration_), "Frame2Model::iteration_");
    /* the pose history stays on the device until history() is called (the reference's caller only draws it,
     * SurfelMapping.cpp:391): the minimisation costs the host one poll of a pinned record, no copy */
    check(ctx_.get(), suma_icp_minimize(ctx_.get(), T0, pose_, nullptr, 0, &n_hist_, &stats_), "LieGaussNewton::minimize");
    history_valid_ = false;
    history_seq_ = suma_icp_history_sequence(ctx_.get());
    std::memcpy(F.pose_, pose_, sizeof(pose_));
    F.stats_ = stats_;
    F.iteration_ += stats_.iterations + (stats_.converged ? 1u : 0u); /* one Objective::increment per step, Objective.h:45-48 */
    check(ctx_.get(), suma_icp_information(ctx_.get(), information_), "LieGaussNewton::information");
    return 0; /* the reference returns 0 from every path of minimize (LieGaussNewton.cpp:37) */
  }
  const double* pose() const { return pose_; }
  double residual() const { return stats_.error; }
  /* LieGaussNewton::reason (LieGaussNewton.cpp:110-115), called by SurfelMapping.cpp:428,448 */
  std::string reason(int32_t errorno) const {
    if (errorno == -1) return "Maximum number of iterations reached.";
    if (errorno == -2) return "Diverging.";
    return "no error";
  }
  /* information(): J^T W J of the last step, 6x6 column-major (LieGaussNewton.cpp:75,103-105) */
  const double* information() const { return information_; }
  /* 16 doubles per entry; fetched from the device on first use after a minimisation */
  const std::vector<double>& history() {
    if (!history_valid_) {
      /* one history buffer per context: another optimizer object (or a loop-closure verification) that has minimised
       * since would be read here instead of this object's chain */
      if (suma_icp_history_sequence(ctx_.get()) != history_seq_)
        throw std::runtime_error("LieGaussNewton::history: a later minimisation on this context has overwritten the "
                                 "device-side history; call history() before it");
      const uint32_t n = n_hist_ < 1025 ? n_hist_ : 1025;
      history_.assign(16 * (size_t)n, 0.0);
      if (n) check(ctx_.get(), suma_icp_history(ctx_.get(), history_.data(), n, nullptr), "LieGaussNewton::history");
      history_valid_ = true;
    }
    return history_;
  }
  uint32_t iterationCount() const { return stats_.iterations; }

 private:
  Context& ctx_;
  double pose_[16];
  double information_[36];
  std::vector<double> history_;
  uint32_t n_hist_{0};
  bool history_valid_{true};
  uint64_t history_seq_{0};
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
  /* getModelSurfels() / getDataSurfels() (SurfelMap.h:64-67) return glow::GlBuffer handles in the reference; here the
   * device buffer of the active map (64-byte records, valid until the next update), for hipGraphicsGLRegisterBuffer
   * or a device-to-device copy into the viewer's VBO.  The data surfels are the tail [first, first + n). */
  struct DeviceSurfels {
    const suma_surfel* d_ptr;
    uint32_t first, n;
  };
  DeviceSurfels getModelSurfels() {
    void* p = nullptr;
    uint32_t n = 0;
    check(ctx_.get(), suma_map_export_surfels(ctx_.get(), &p, &n), "SurfelMap::getModelSurfels");
    return DeviceSurfels{(const suma_surfel*)p, 0, n};
  }
  DeviceSurfels getDataSurfels() {
    void* p = nullptr;
    uint32_t first = 0, n = 0;
    check(ctx_.get(), suma_map_export_data_surfels(ctx_.get(), &p, &first, &n), "SurfelMap::getDataSurfels");
    return DeviceSurfels{(const suma_surfel*)p, first, n};
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

/* src/core/SurfelMapping.h: the per-scan sequencing (SurfelMapping.cpp:175-210) on the scan pipeline of the library --
 * K1-K3 on a side stream, the Gauss-Newton chain resident on the device, K7 / K8 fused into the passes around them,
 * renders de-duplicated.  processScan keeps the reference's shape: integrateLoopClosures and checkLoopClosure stay the
 * maintainer's functions (pose graph, candidate search, gtsam: untouched) and call the hooks below for their device
 * parts.  Poses cross as column-major double[16] (Eigen::Matrix4d::data()). */
class SurfelMapping {
 public:
  explicit SurfelMapping(const suma_params& p, int device = 0) {
    if (suma_pipeline_create(&p, device, &s_) != SUMA_OK)
      throw std::runtime_error(std::string("suma_pipeline_create: ") + suma_last_error(nullptr));
  }
  ~SurfelMapping() { suma_pipeline_destroy(s_); }
  SurfelMapping(const SurfelMapping&) = delete;
  SurfelMapping& operator=(const SurfelMapping&) = delete;

  /* processScan(const rv::Laserscan&), SurfelMapping.cpp:175-210.  The two std::function hooks are where the
   * reference calls integrateLoopClosures() (:179) and checkLoopClosure() (:196); leave them empty for
   * close-loops = false.  fixed_iterations = 0: the stopping tests of LieGaussNewton decide. */
  template <class Before, class Between>
  void processScan(const suma_float4* points, const float* labels, const float* probs, uint32_t n, Before integrate,
                   Between check_loop_closure, int32_t fixed_iterations = 0) {
    const double t_all = now();
    integrate(*this);                                                                            /* :179 */
    double t = now();
    chk(suma_pipeline_begin_scan(s_, points, labels, probs, n), "SurfelMapping::initialize/preprocess"); /* :181-187 */
    statistics_["initialize-time"] = 0.0; /* the frame swaps of initialize() (:323-331) are pointer swaps inside begin_scan */
    statistics_["preprocessing-time"] = now() - t;                                              /* :187 */
    const bool tracked = timestamp() > 0;
    t = now();
    chk(suma_pipeline_update_pose(s_, fixed_iterations), "SurfelMapping::updatePose");          /* :192 */
    if (tracked) {
      const double dt = now() - t;
      statistics_["icp-time"] = dt;                                                             /* :193 (and :425) */
      statistics_["opt-time"] = dt;                                                             /* :393 */
      statistics_["icp-overall"] = dt;                                                          /* :475 */
      suma_icp_stats st;
      std::memset(&st, 0, sizeof(st));
      suma_pipeline_minimize_stats(s_, &st);
      statistics_["num_iterations"] = (double)st.iterations;                                    /* :394 */
      t = now();
      check_loop_closure(*this);                                                                /* :196 */
      statistics_["loop-time"] = now() - t;                                                     /* :197 */
    }
    t = now();
    chk(suma_pipeline_update_map(s_), "SurfelMapping::updateMap");                               /* :201, :209 */
    const double dt_map = now() - t;
    statistics_["map-update"] = dt_map;                                                          /* :800 */
    statistics_["mapping-time"] = dt_map;                                                        /* :202 */
    const double complete = now() - t_all;
    statistics_["complete-time"] = complete;                                                     /* :206 */
    statistics_["icp_percentage"] = complete > 0.0 ? statistics_["opt-time"] / complete : 0.0;  /* :207 */
  }
  void processScan(const suma_float4* points, const float* labels, const float* probs, uint32_t n,
                   int32_t fixed_iterations = 0) {
    auto nop = [](SurfelMapping&) {};
    processScan(points, labels, probs, n, nop, nop, fixed_iterations);
  }
  /* SurfelMapping::Stats (SurfelMapping.h; filled at SurfelMapping.cpp:183-207, 393-394, 425, 475, 800): the keys the
   * untouched GUI plots (VisualizerWindow.cpp:705), in seconds like rv::Stopwatch::toc().  They are HOST clocks around
   * the phase calls, as the reference's are around its GL calls -- with one difference in meaning: the reference
   * drains the GL pipeline inside every phase (glFinish), this library only waits where the host needs a result (the
   * minimisation), so "mapping-time" is the time to ENQUEUE the update and the post-update rendering, whose GPU work
   * overlaps the next scan's upload.  Per-kernel GPU times: gpuTimes(). */
  const std::map<std::string, double>& getStatistics() const { return statistics_; }
  /* GPU time per kernel group from suma_profile_get (ms summed since the last reset; needs suma_profile_enable(ctx, 1),
   * which costs a few percent of throughput) */
  std::map<std::string, double> gpuTimes() {
    std::map<std::string, double> out;
    suma_kernel_time kt[64];
    const int n = suma_profile_get(ctx(), kt, 64);
    for (int i = 0; i < n && i < 64; ++i) out[kt[i].name] = kt[i].total_ms;
    return out;
  }
  /* SurfelMapping::reset(), SurfelMapping.cpp:131-169 */
  void reset() { chk(suma_pipeline_reset(s_), "SurfelMapping::reset"); }
  /* ---- device parts of checkLoopClosure ---- */
  /* :546-574, a tracked closure verified again; on success the caller sets currentPose_old_ = out.pose_old (:581) */
  suma_loop_track trackLoopClosure(double min_valid = 0.2, double max_outlier = 0.85, double max_diff = 0.1) {
    suma_loop_track t;
    chk(suma_pipeline_track_loop_closure(s_, min_valid, max_outlier, max_diff, &t), "checkLoopClosure (tracking)");
    return t;
  }
  /* :679-757, a candidate verified from n_init initial guesses (16 doubles each) */
  std::vector<suma_loop_result> verifyLoopClosure(const double* pose_prior, const double* initializations,
                                                  uint32_t n_init, float min_valid = 0.2f, float max_outlier = 0.85f) {
    std::vector<suma_loop_result> out(n_init);
    chk(suma_pipeline_verify_loop_closure(s_, pose_prior, initializations, n_init, min_valid, max_outlier, out.data()),
        "checkLoopClosure (candidates)");
    return out;
  }
  void setCurrentPoseOld(const double* pose_old) { chk(suma_pipeline_set_pose_old(s_, pose_old), "currentPose_old_"); }
  /* ---- integrateLoopClosures, :211-250: casted_poses (16 floats each), difference ---- */
  void integrateLoopClosures(const std::vector<float>& casted_poses, const double* difference) {
    chk(suma_pipeline_integrate_loop_closures(s_, casted_poses.data(), (uint32_t)(casted_poses.size() / 16), difference),
        "SurfelMapping::integrateLoopClosures");
  }
  /* which: 0 currentPose_, 1 currentPose_old_, 2 currentPose_new_, 3 lastPose_old_, 4 lastPose_ */
  void getPose(int which, double* pose16) const { suma_pipeline_get_pose(s_, which, pose16); }
  void getCurrentPose(double* pose16) const { suma_pipeline_pose(s_, pose16); }
  suma_icp_stats resultNew() {
    suma_icp_stats st;
    chk(suma_pipeline_result_new(s_, &st), "result_new_");
    return st;
  }
  uint32_t timestamp() const { return suma_pipeline_timestamp(s_); }
  uint32_t trackLoss() const { return suma_pipeline_track_loss(s_); }
  suma_pipeline* get() const { return s_; }
  suma_ctx* ctx() const { return suma_pipeline_ctx(s_); }

 private:
  void chk(int rc, const char* what) const { check(suma_pipeline_ctx(s_), rc, what); }
  static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  suma_pipeline* s_{nullptr};
  std::map<std::string, double> statistics_;
};

}  // namespace suma_hip
#endif
