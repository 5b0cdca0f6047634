/*
 * tests/cpp/ref_shells/ref_shells.cpp -- the method bodies a maintainer of PRBonn/semantic_suma puts behind the
 * reference's class interfaces (core_shells.h: the EXACT signatures of src/core/Preprocessing.h:47-58, Objective.h:14-82,
 * Frame2Model.h:28-52, LieGaussNewton.h:25-58, SurfelMap.h:36-78) so that an unchanged SurfelMapping.cpp drives
 * libsuma_hip.so: each body is a few lines over include/suma_adapter.hpp.  This file is INTEGRATION.md section 2 as
 * code that the test suite compiles (CPU) and runs against the scan pipeline (GPU).
 */
#include "core_shells.h"

#include <cmath>
#include <cstring>

namespace suma_shell {

/* rv::ParameterList -> suma_params: the keys of SURVEY.md Appendix C, read the way the reference's constructors and
 * setParameters() read them (required keys throw when missing, optional ones keep the reference's defaults):
 * Preprocessing.cpp:25-26,77-100; Frame2Model.cpp:65-110; LieGaussNewton.cpp:81-91; SurfelMap.cpp:8-13,265-273,336-449;
 * SurfelMapping.cpp:73-113. */
suma_params params_from(const rv::ParameterList& params) {
  suma_params p;
  suma_params_default(&p);
  p.data_width = params["data_width"];
  p.data_height = params["data_height"];
  p.data_fov_up = params["data_fov_up"];
  p.data_fov_down = params["data_fov_down"];
  p.min_depth = params["min_depth"];
  p.max_depth = params["max_depth"];
  p.model_width = params["model_width"];
  p.model_height = params["model_height"];
  p.model_fov_up = params["model_fov_up"];
  p.model_fov_down = params["model_fov_down"];
  p.model_min_depth = params["model_min_depth"];
  p.model_max_depth = params["model_max_depth"];
  p.max_iterations = params["max iterations"];
  p.stopping_threshold = params["stopping threshold"];
  p.delta = params["delta"];
  p.icp_max_distance = params["icp-max-distance"];
  p.icp_max_angle = params["icp-max-angle"];
  p.weight_function = SUMA_WEIGHT_NONE; /* Frame2Model.cpp:69-80 */
  p.factor = 1.0f;
  if (params.hasParam("weighting")) {
    const std::string w = params["weighting"];
    if (w == "huber") p.weight_function = SUMA_WEIGHT_HUBER;
    else if (w == "turkey") p.weight_function = SUMA_WEIGHT_TUKEY;
    else if (w == "stability") p.weight_function = SUMA_WEIGHT_STABILITY;
    p.factor = params["factor"];
  }
  p.bilinear_sampling = (params.hasParam("bilinear_sampling") && (bool)params["bilinear_sampling"]) ? 1 : 0;
  p.initialize_identity = (bool)params["initialize_identity"] ? 1 : 0;
  p.fallback_mode = (bool)params["fallback_mode"] ? 1 : 0;
  if (p.fallback_mode) {
    p.fallback_max_distance = params["fallback-max-distance"];
    p.fallback_max_angle = params["fallback-max-angle"];
  }
  p.compose_rendering = (bool)params["compose_rendering"] ? 1 : 0;
  p.max_loop_closure_distance = params["max_loop_closure_distance"];
  if (params.hasParam("min_radius")) p.min_radius = params["min_radius"];
  if (params.hasParam("max_radius")) p.max_radius = params["max_radius"];
  if (params.hasParam("max_angle")) p.max_angle = params["max_angle"];
  p.map_max_distance = params["map-max-distance"];
  p.map_max_angle = params["map-max-angle"];
  p.unstable_age = params["unstable_age"];
  p.confidence_mode = params["confidence_mode"];
  p.confidence_threshold = params["confidence_threshold"];
  p.p_stable = params["p_stable"];
  p.p_prior = params["p_prior"];
  p.sigma_angle = params["sigma_angle"];
  p.sigma_distance = params["sigma_distance"];
  if (params.hasParam("use_stability")) p.use_stability = (bool)params["use_stability"] ? 1 : 0;
  if (params.hasParam("active_timestamps")) p.active_timestamps = params["active_timestamps"];
  if (params.hasParam("max_weight")) p.max_weight = params["max_weight"];
  if (params.hasParam("weighting_scheme")) p.weighting_scheme = params["weighting_scheme"];
  if (params.hasParam("averaging_scheme")) p.averaging_scheme = params["averaging_scheme"];
  if (params.hasParam("update_always")) p.update_always = (bool)params["update_always"] ? 1 : 0;
  p.submap_dimension = params["submap-dimension"];
  p.submap_extent = params["submap-extent"];
  if (params.hasParam("partial-extraction")) p.partial_extraction = (bool)params["partial-extraction"] ? 1 : 0;
  p.filter_vertexmap = p.avg_vertexmap = p.use_filtered_vertexmap = 0; /* Preprocessing.cpp:81-90 */
  if (params.hasParam("filter_vertexmap")) p.filter_vertexmap = (bool)params["filter_vertexmap"] ? 1 : 0;
  if (params.hasParam("avg_vertexmap")) p.avg_vertexmap = (bool)params["avg_vertexmap"] ? 1 : 0;
  if (p.filter_vertexmap) {
    p.bilateral_sigma_space = params["bilateral_sigma_space"];
    p.bilateral_sigma_range = params["bilateral_sigma_range"];
    p.use_filtered_vertexmap = (bool)params["use_filtered_vertexmap"] ? 1 : 0;
  }
  return p;
}

static std::unique_ptr<suma_hip::Context>& slot() {
  static std::unique_ptr<suma_hip::Context> c;
  return c;
}
suma_hip::Context& context(const rv::ParameterList& params) {
  if (!slot()) slot().reset(new suma_hip::Context(params_from(params), 0));
  return *slot();
}
suma_hip::Context& context() {
  if (!slot()) throw std::runtime_error("suma_shell::context: no hot-path object has been constructed yet");
  return *slot();
}

}  // namespace suma_shell

/* ---- Frame (src/core/Frame.h:21-79) ---- */
Frame::Frame(uint32_t w, uint32_t h)
    : valid(false), width(w), height(h), hip(std::make_shared<suma_hip::Frame>(suma_shell::context(), w, h)) {
  points.reserve(150000);
}
void Frame::copy(const Frame& other) { /* Frame.h:49-61 */
  valid = other.valid;
  hip->copy(*other.hip); /* vertex_map, normal_map, semantic_map */
  points.assign(other.points);
  map = other.map;
  pose = other.pose;
}

/* ---- Preprocessing (src/core/Preprocessing.cpp:16-118, 120-339) ---- */
Preprocessing::Preprocessing(const rv::ParameterList& params) : hip_(suma_shell::context(params)) {}
void Preprocessing::setParameters(const rv::ParameterList& params) {
  suma_shell::context().setParameters(suma_shell::params_from(params));
}
void Preprocessing::process(glow::GlBuffer<rv::Point3f>& points, Frame& frame, glow::GlBuffer<float>& labels,
                            glow::GlBuffer<float>& probs, uint32_t timestamp_) {
  frame.points.assign(points); /* Preprocessing.cpp:123-125 */
  frame.labels.assign(labels);
  frame.probs.assign(probs);
  std::vector<rv::Point3f> p;
  std::vector<float> l, q;
  points.get(p); /* with real glow: keep the host vectors of SurfelMapping::initialize instead of reading back */
  labels.get(l);
  probs.get(q);
  static_assert(sizeof(rv::Point3f) == sizeof(suma_float4), "rv::Point3f is the 16-byte x, y, z, 1 record");
  hip_.process(reinterpret_cast<const suma_float4*>(p.data()), (uint32_t)p.size(), *frame.hip,
               l.empty() ? nullptr : l.data(), q.empty() ? nullptr : q.data(), timestamp_);
  frame.valid = true;
}

/* ---- Objective / Frame2Model (src/core/Objective.h:45-48, Frame2Model.cpp:14-134, 136-261) ---- */
void Objective::increment(const Eigen::VectorXd& delta) {
  double E[16];
  suma_hip::se3_exp(delta.data(), E);
  Eigen::Matrix4d Em;
  std::memcpy(Em.data(), E, sizeof(E));
  pose_ = Em * pose_;
  iteration_ += 1;
}
Frame2Model::Frame2Model(const rv::ParameterList& params)
    : hip_(suma_shell::context(params), suma_shell::params_from(params)) {}
void Frame2Model::setParameter(const rv::Parameter& param) { /* Frame2Model.cpp:112-115 is empty; the keys it owns: */
  if (param.isString()) hip_.setParameter(param.name(), (std::string)param);
  else hip_.setParameter(param.name(), (double)param);
}
void Frame2Model::setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last) {
  current_ = current; /* Frame2Model.cpp:117-123 */
  last_ = last;
  iteration_ = 0;
  hip_.setData(current->hip, last->hip);
}
void Frame2Model::setLevel(uint32_t) {}
uint32_t Frame2Model::getMaxLevel() const { return 0; }
uint32_t Frame2Model::num_parameters() const { return 6; }
double Frame2Model::residual(const Eigen::VectorXd&) { throw std::runtime_error("not implemented."); } /* :131-134 */
double Frame2Model::jacobianProducts(Eigen::MatrixXd& JtJ, Eigen::MatrixXd& Jtf) {
  JtJ.resize(6, 6);
  Jtf.resize(6, 1);
  hip_.initialize(pose_.data()); /* resets the adapter's counter: hand ours over again (Tukey reads it) */
  for (uint32_t i = 0; i < iteration_; ++i) hip_.increment((const double[6]){0, 0, 0, 0, 0, 0});
  const double F = hip_.jacobianProducts(JtJ.data(), Jtf.data());
  inlier_ = hip_.inlier(); /* Frame2Model.cpp:222-227 */
  outlier_ = hip_.outlier();
  invalid_ = hip_.invalid();
  inlier_residual_ = (float)hip_.inlier_residual();
  return F;
}

/* ---- LieGaussNewton (src/core/LieGaussNewton.cpp:13-115): the loop runs on the device ---- */
LieGaussNewton::LieGaussNewton() : hip_(suma_shell::context()) {}
void LieGaussNewton::setParameters(const rv::ParameterList& params) { /* :81-91: max iterations, stopping threshold, delta */
  suma_params p = suma_shell::context().params();
  if (params.hasParam("max iterations")) p.max_iterations = params["max iterations"];
  if (params.hasParam("stopping threshold")) p.stopping_threshold = params["stopping threshold"];
  if (params.hasParam("delta")) p.delta = params["delta"];
  suma_shell::context().setParameters(p);
}
int32_t LieGaussNewton::minimize(Objective& F, const Eigen::Matrix4d& T0) {
  Frame2Model& f2m = static_cast<Frame2Model&>(F); /* the only Objective of the hot path (SurfelMapping.cpp:84-94) */
  hip_.minimize(f2m.hip_, T0.data());
  std::memcpy(Tk_.data(), hip_.pose(), 16 * sizeof(double));
  f2m.pose_ = Tk_;
  f2m.iteration_ = hip_.iterationCount();
  f2m.inlier_ = f2m.hip_.inlier();
  f2m.outlier_ = f2m.hip_.outlier();
  f2m.invalid_ = f2m.hip_.invalid();
  f2m.inlier_residual_ = (float)f2m.hip_.inlier_residual();
  information_.resize(6, 6);
  std::memcpy(information_.data(), hip_.information(), 36 * sizeof(double));
  history_fetched_ = false;
  return 0; /* LieGaussNewton.cpp:37 */
}
double LieGaussNewton::residual() const { return hip_.residual(); }
const Eigen::Matrix4d& LieGaussNewton::pose() const { return Tk_; }
std::string LieGaussNewton::reason(int32_t errorno) const { return hip_.reason(errorno); }
const Eigen::MatrixXd& LieGaussNewton::information() { return information_; }
uint32_t LieGaussNewton::iterationCount() const { return hip_.iterationCount(); }
const std::vector<Eigen::Matrix4d>& LieGaussNewton::history() const {
  if (!history_fetched_) {
    const std::vector<double>& h = hip_.history();
    history_.assign(h.size() / 16, Eigen::Matrix4d());
    for (size_t k = 0; k < history_.size(); ++k) std::memcpy(history_[k].data(), h.data() + 16 * k, 16 * sizeof(double));
    history_fetched_ = true;
  }
  return history_;
}

/* ---- SurfelMap (src/core/SurfelMap.cpp:8-457, 473-584, 847-1165, 1232-1237) ---- */
static std::shared_ptr<Frame> wrap(const suma_params& p, const std::shared_ptr<suma_hip::Frame>& f) {
  std::shared_ptr<Frame> out = std::make_shared<Frame>(p.model_width, p.model_height);
  out->hip = f; /* the map's own frame, not a copy */
  return out;
}
SurfelMap::SurfelMap(const rv::ParameterList& params) : hip_(suma_shell::context(params)) {
  const suma_params& p = suma_shell::context().params();
  oldMapFrame_ = wrap(p, hip_.oldMapFrame()); /* SurfelMap.cpp:451-453 */
  newMapFrame_ = wrap(p, hip_.newMapFrame());
  composedFrame_ = wrap(p, hip_.composedFrame());
}
void SurfelMap::setParameters(const rv::ParameterList& params) {
  suma_shell::context().setParameters(suma_shell::params_from(params));
}
void SurfelMap::reset() { hip_.reset(); }
void SurfelMap::update(const Eigen::Matrix4f& pose, Frame& frame) { hip_.update(pose.data(), *frame.hip); }
void SurfelMap::render(const Eigen::Matrix4f& pose, Frame& frame, float confidence_threshold) {
  hip_.render(pose.data(), *frame.hip, confidence_threshold);
}
void SurfelMap::render(const Eigen::Matrix4f& pose_old, const Eigen::Matrix4f& pose_new, Frame& frame,
                       float confidence_threshold) {
  hip_.render(pose_old.data(), pose_new.data(), *frame.hip, confidence_threshold);
}
void SurfelMap::render_active(const Eigen::Matrix4f& pose, float confidence_threshold) {
  hip_.render_active(pose.data(), confidence_threshold);
}
void SurfelMap::render_inactive(const Eigen::Matrix4f& pose, float confidence_threshold) {
  hip_.render_inactive(pose.data(), confidence_threshold);
}
void SurfelMap::render_composed(const Eigen::Matrix4f& pose_old, const Eigen::Matrix4f& pose_new,
                                float confidence_threshold) {
  hip_.render_composed(pose_old.data(), pose_new.data(), confidence_threshold);
}
std::shared_ptr<Frame>& SurfelMap::oldMapFrame() { return oldMapFrame_; }
std::shared_ptr<Frame>& SurfelMap::newMapFrame() { return newMapFrame_; }
std::shared_ptr<Frame>& SurfelMap::composedFrame() { return composedFrame_; }
uint32_t SurfelMap::size() const { return hip_.size(); }
void SurfelMap::updatePoses(const std::vector<Eigen::Matrix4f>& poses) {
  std::vector<float> flat(16 * poses.size());
  for (size_t k = 0; k < poses.size(); ++k) std::memcpy(flat.data() + 16 * k, poses[k].data(), 16 * sizeof(float));
  hip_.updatePoses(flat);
}
std::vector<Surfel> SurfelMap::getAllSurfels() { return hip_.getAllSurfels(); }
