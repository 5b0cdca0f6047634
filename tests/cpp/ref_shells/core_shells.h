/*
 * tests/cpp/ref_shells/core_shells.h -- the hot-path classes of PRBonn/semantic_suma's src/core declared with the
 * reference's EXACT public signatures, as a maintainer's tree would have them after the GL members are replaced by one
 * `hip_` member each.  ref_shells.cpp defines the methods over include/suma_adapter.hpp.
 *
 *   Frame            src/core/Frame.h:21-79          (members the callers touch: width, height, valid, pose, points, labels, probs)
 *   Preprocessing    src/core/Preprocessing.h:47-58
 *   Objective        src/core/Objective.h:14-82
 *   Frame2Model      src/core/Frame2Model.h:28-52
 *   LieGaussNewton   src/core/LieGaussNewton.h:25-58
 *   SurfelMap        src/core/SurfelMap.h:36-78      (draw / setColorMap are visualisation: out of scope, left out)
 *
 * tests/test_ref_shells.py compiles this against stub glow / rv / Eigen headers (stubs/), and -- where /root/reference is
 * present -- checks every signature below against the text of the reference's own headers, so the prose of
 * INTEGRATION.md cannot rot.  Nothing here is shipped: the product boundary is include/suma_hip.h.
 */
#ifndef REF_SHELLS_CORE_SHELLS_H_
#define REF_SHELLS_CORE_SHELLS_H_

#include <glow/GlBuffer.h>
#include <rv/ParameterList.h>
#include <rv/geometry.h>
#include <eigen3/Eigen/Dense>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "suma_adapter.hpp"

/* the one device context of the process, created from the first rv::ParameterList a hot-path class is built with -- the
 * reference shares one GL context between the same objects (visualizer.cpp:19) */
namespace suma_shell {
suma_params params_from(const rv::ParameterList& params);  /* Appendix-C keys of config/default.xml -> suma_params */
suma_hip::Context& context(const rv::ParameterList& params);
suma_hip::Context& context();
}  // namespace suma_shell

typedef suma_surfel Surfel; /* src/core/Surfel.h:5-15: the same 64-byte record */

class SurfelMap;

class Frame {
 public:
  typedef std::shared_ptr<Frame> Ptr;
  Frame(uint32_t w, uint32_t h);
  void copy(const Frame& other);

  bool valid;
  uint32_t width, height;
  glow::GlBuffer<rv::Point3f> points{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};
  glow::GlBuffer<float> labels{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};
  glow::GlBuffer<float> probs{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};
  Eigen::Matrix4f pose{Eigen::Matrix4f::Identity()};
  std::shared_ptr<SurfelMap> map;

  std::shared_ptr<suma_hip::Frame> hip; /* vertex_map / normal_map / semantic_map live here (suma_frame) */
};

class Preprocessing {
 public:
  Preprocessing(const rv::ParameterList& params);
  void setParameters(const rv::ParameterList& params);
  void process(glow::GlBuffer<rv::Point3f>& points, Frame& frame, glow::GlBuffer<float>& labels,
               glow::GlBuffer<float>& probs, uint32_t timestamp_);

 protected:
  suma_hip::Preprocessing hip_;
};

class Objective {
 public:
  virtual ~Objective() {}
  virtual uint32_t num_parameters() const = 0;
  virtual void setParameter(const rv::Parameter& param) {}
  virtual void setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last) {
    throw std::runtime_error("ProjectiveICP::setData not implemented.");
  }
  virtual double residual(const Eigen::VectorXd& delta) = 0;
  virtual double jacobianProducts(Eigen::MatrixXd& JtJ, Eigen::MatrixXd& Jtf) = 0;
  void increment(const Eigen::VectorXd& delta);
  uint32_t inlier() const { return inlier_; }
  uint32_t outlier() const { return outlier_; }
  uint32_t valid() const { return (inlier_ + outlier_); }
  uint32_t invalid() const { return invalid_; }
  float inlier_residual() const { return inlier_residual_; }
  void initialize(const Eigen::Matrix4d& T0) { pose_ = T0; }
  const Eigen::Matrix4d& pose() const { return pose_; }
  virtual void setLevel(uint32_t lvl) {}
  virtual uint32_t getMaxLevel() const { return 0; }
  virtual void reset() {}

 protected:
  Objective() {}
  Eigen::Matrix4d pose_{Eigen::Matrix4d::Identity()};
  uint32_t iteration_{0};
  uint32_t inlier_{0};
  float inlier_residual_{0.0f};
  uint32_t outlier_{0};
  uint32_t invalid_{0};
};

class LieGaussNewton;

class Frame2Model : public Objective {
 public:
  Frame2Model(const rv::ParameterList& params);
  void setParameter(const rv::Parameter& param) override;
  void setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last);
  void setLevel(uint32_t lvl) override;
  uint32_t getMaxLevel() const override;
  uint32_t num_parameters() const;
  double residual(const Eigen::VectorXd& delta);
  double jacobianProducts(Eigen::MatrixXd& JtJ, Eigen::MatrixXd& Jtf);

 protected:
  friend class LieGaussNewton;
  suma_hip::Frame2Model hip_;
  std::shared_ptr<Frame> current_;
  std::shared_ptr<Frame> last_;
};

class LieGaussNewton {
 public:
  LieGaussNewton();
  void setParameters(const rv::ParameterList& params);
  int32_t minimize(Objective& F, const Eigen::Matrix4d& T0);
  double residual() const;
  const Eigen::Matrix4d& pose() const;
  std::string reason(int32_t errorno) const;
  const Eigen::MatrixXd& information();
  uint32_t iterationCount() const;
  static const int32_t CONVERGED{0};
  const std::vector<Eigen::Matrix4d>& history() const;

 protected:
  mutable suma_hip::LieGaussNewton hip_;
  Eigen::Matrix4d Tk_;
  Eigen::MatrixXd information_;
  mutable std::vector<Eigen::Matrix4d> history_; /* fetched from the device on first use after a minimisation */
  mutable bool history_fetched_{true};
};

class SurfelMap {
 public:
  SurfelMap(const rv::ParameterList& params);
  void setParameters(const rv::ParameterList& params);
  void reset();
  void update(const Eigen::Matrix4f& pose, Frame& frame);
  void render(const Eigen::Matrix4f& pose, Frame& frame, float confidence_threshold);
  void render(const Eigen::Matrix4f& pose_old, const Eigen::Matrix4f& pose_new, Frame& frame,
              float confidence_threshold);
  void render_active(const Eigen::Matrix4f& pose, float confidence_threshold);
  void render_inactive(const Eigen::Matrix4f& pose, float confidence_threshold);
  void render_composed(const Eigen::Matrix4f& pose_old, const Eigen::Matrix4f& pose_new, float confidence_threshold);
  std::shared_ptr<Frame>& oldMapFrame();
  std::shared_ptr<Frame>& newMapFrame();
  std::shared_ptr<Frame>& composedFrame();
  uint32_t size() const;
  void updatePoses(const std::vector<Eigen::Matrix4f>& poses);
  std::vector<Surfel> getAllSurfels();

 protected:
  mutable suma_hip::SurfelMap hip_;
  std::shared_ptr<Frame> oldMapFrame_, newMapFrame_, composedFrame_;
};

#endif
