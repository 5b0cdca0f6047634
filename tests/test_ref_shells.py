"""The reference-side binding of INTEGRATION.md section 2 as code that cannot rot: tests/cpp/ref_shells/ declares the
hot-path classes with the reference's EXACT public signatures (src/core/Preprocessing.h:47-58, Objective.h:14-82,
Frame2Model.h:28-52, LieGaussNewton.h:25-58, SurfelMap.h:36-78), defines every method over include/suma_adapter.hpp, and
drives them with SurfelMapping's own call sequence.  CPU: it compiles and links against libsuma_hip.so with stub
glow / rv / Eigen headers, the rv::ParameterList -> suma_params filler reproduces config/default.xml, and (where
/root/reference exists) every signature is found verbatim in the reference's header.  GPU: the call sequence produces
the scan pipeline's pose bits."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import get_scan
from semantic_suma_amd.types import params_with_size

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHELLS = os.path.join(ROOT, "tests", "cpp", "ref_shells")
REF_CORE = "/root/reference/src/core"

# (reference header, signatures that core_shells.h must share with it, word for word)
SIGNATURES = {
    "Preprocessing.h": [
        "Preprocessing(const rv::ParameterList& params);",
        "void setParameters(const rv::ParameterList& params);",
        "void process(glow::GlBuffer<rv::Point3f>& points, Frame& frame, glow::GlBuffer<float>& labels, glow::GlBuffer<float>& probs, uint32_t timestamp_);",
    ],
    "Objective.h": [
        "virtual uint32_t num_parameters() const = 0;",
        "virtual void setParameter(const rv::Parameter& param) {}",
        "virtual void setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last) {",
        "virtual double residual(const Eigen::VectorXd& delta) = 0;",
        "virtual double jacobianProducts(Eigen::MatrixXd& JtJ, Eigen::MatrixXd& Jtf) = 0;",
        "void increment(const Eigen::VectorXd& delta)",
        "uint32_t inlier() const { return inlier_; }",
        "uint32_t valid() const { return (inlier_ + outlier_); }",
        "float inlier_residual() const { return inlier_residual_; }",
        "void initialize(const Eigen::Matrix4d& T0) { pose_ = T0; }",
        "const Eigen::Matrix4d& pose() const { return pose_; }",
        "virtual uint32_t getMaxLevel() const { return 0; }",
    ],
    "Frame2Model.h": [
        "class Frame2Model : public Objective {",
        "Frame2Model(const rv::ParameterList& params);",
        "void setParameter(const rv::Parameter& param) override;",
        "void setData(const std::shared_ptr<Frame>& current, const std::shared_ptr<Frame>& last);",
        "void setLevel(uint32_t lvl) override;",
        "uint32_t getMaxLevel() const override;",
        "uint32_t num_parameters() const;",
        "double residual(const Eigen::VectorXd& delta);",
        "double jacobianProducts(Eigen::MatrixXd& JtJ, Eigen::MatrixXd& Jtf);",
    ],
    "LieGaussNewton.h": [
        "LieGaussNewton();",
        "void setParameters(const rv::ParameterList& params);",
        "int32_t minimize(Objective& F, const Eigen::Matrix4d& T0);",
        "double residual() const;",
        "const Eigen::Matrix4d& pose() const;",
        "std::string reason(int32_t errorno) const;",
        "const Eigen::MatrixXd& information();",
        "uint32_t iterationCount() const;",
        "static const int32_t CONVERGED{0};",
        "const std::vector<Eigen::Matrix4d>& history() const",
    ],
    "SurfelMap.h": [
        "SurfelMap(const rv::ParameterList& params);",
        "void setParameters(const rv::ParameterList& params);",
        "void reset();",
        "void update(const Eigen::Matrix4f& pose, Frame& frame);",
        "void render(const Eigen::Matrix4f& pose, Frame& frame, float confidence_threshold);",
        "void render(const Eigen::Matrix4f& pose_old, const Eigen::Matrix4f& pose_new, Frame& frame, float confidence_threshold);",
        "void render_active(const Eigen::Matrix4f& pose, float confidence_threshold);",
        "void render_inactive(const Eigen::Matrix4f& pose, float confidence_threshold);",
        "void render_composed(const Eigen::Matrix4f& pose_old, const Eigen::Matrix4f& pose_new, float confidence_threshold);",
        "std::shared_ptr<Frame>& oldMapFrame();",
        "std::shared_ptr<Frame>& newMapFrame();",
        "std::shared_ptr<Frame>& composedFrame();",
        "void updatePoses(const std::vector<Eigen::Matrix4f>& poses);",
        "std::vector<Surfel> getAllSurfels();",
    ],
    "Frame.h": [
        "typedef std::shared_ptr<Frame> Ptr;",
        "Frame(uint32_t w, uint32_t h)",
        "void copy(const Frame& other)",
        "uint32_t width, height;",
        "glow::GlBuffer<rv::Point3f> points{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};",
        "glow::GlBuffer<float> labels{glow::BufferTarget::ARRAY_BUFFER, glow::BufferUsage::DYNAMIC_DRAW};",
        "Eigen::Matrix4f pose{Eigen::Matrix4f::Identity()};",
    ],
}


def squash(text):
    """comments out, all whitespace out: a declaration wrapped over two lines equals its one-line form"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def build_driver(tmp):
    from semantic_suma_amd import core
    libdir = os.path.dirname(core.LIB_PATH)
    exe = os.path.join(tmp, "ref_shells_driver")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror=return-type", "-I", os.path.join(SHELLS, "stubs"), "-I", SHELLS,
                           "-I", os.path.join(ROOT, "include"), os.path.join(SHELLS, "ref_shells.cpp"),
                           os.path.join(SHELLS, "ref_shells_driver.cpp"), "-o", exe, "-L", libdir, "-lsuma_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shell_header_declares_the_reference_signatures():
    """every signature of the list is in core_shells.h -- and, where the reference tree is present, in the reference's own
    header, word for word (whitespace and comments apart)"""
    mine = squash(open(os.path.join(SHELLS, "core_shells.h")).read())
    for header, sigs in SIGNATURES.items():
        ref = squash(open(os.path.join(REF_CORE, header)).read()) if os.path.isdir(REF_CORE) else None
        for s in sigs:
            assert squash(s) in mine, f"core_shells.h lacks `{s}`"
            if ref is not None:
                assert squash(s) in ref, f"{header} of the reference has no `{s}`"


def test_shells_compile_link_and_fill_suma_params_from_a_parameter_list(tmp_path):
    """g++ against the stub headers + libsuma_hip.so (no GPU needed to link or to run --params): the rv::ParameterList of
    config/default.xml, read key by key the way the reference's constructors read it, gives suma_params_default()"""
    exe = build_driver(str(tmp_path))
    out = subprocess.check_output([exe, "--params"], timeout=60).decode().strip().splitlines()
    assert len(out) == 2 and out[0] == out[1], "\n".join(out)
    assert out[0].split()[:2] == ["900", "64"] and len(out[0].split()) == 50


def test_default_xml_of_the_driver_is_the_references(tmp_path):
    """the driver's default_xml() against config/default.xml itself (where the reference tree is present)"""
    path = "/root/reference/config/default.xml"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    import xml.etree.ElementTree as ET
    src = open(os.path.join(SHELLS, "ref_shells_driver.cpp")).read()
    mine = {}
    for kind, name, value in re.findall(r'rv::(Integer|Float|Boolean|String)Parameter\("([^"]+)",\s*([^)]+)\)\);', src):
        mine[name] = value.strip().strip('"')
    n = 0
    for el in ET.parse(path).getroot().iter("param"):
        name, typ, text = el.get("name"), el.get("type"), (el.text or "").strip()
        if name not in mine or name in ("data_width", "model_width"):
            continue
        n += 1
        if typ in ("integer", "float"):
            assert float(mine[name].replace("(int)width", "900")) == float(text), name
        elif typ == "boolean":
            assert mine[name] == text, name
        else:
            assert mine[name] == text, name
    assert n >= 45


@pytest.mark.gpu
def test_reference_call_sequence_on_the_shells_gives_the_pipeline_pose_bits(tmp_path):
    from semantic_suma_amd import core as hip
    N, W = 5, 900
    exe = build_driver(str(tmp_path))
    d = tmp_path / "velodyne"
    d.mkdir()
    scans = []
    for k in range(N):
        pts = get_scan(k, W, False)[0].copy()
        pts[:, 3] = 1.0
        pts.astype("<f4").tofile(str(d / f"{k:06d}.bin"))
        scans.append(pts)
    out = subprocess.check_output([exe, str(d), str(N), str(W)], timeout=180).decode().strip().splitlines()
    assert len(out) == N
    pipe = hip.SurfelMapping(params_with_size(W))  # default.xml: 33 iterations at most, the stopping tests decide
    for k in range(N):
        z = np.zeros(scans[k].shape[0], np.float32)
        pipe.processScan(scans[k], z, z, fixed_iterations=0)
        cols = out[k].split()
        got = np.array([struct.unpack("<d", bytes.fromhex(h)[::-1])[0] for h in cols[1:17]]).reshape(4, 4).T
        assert np.array_equal(got, pipe.getCurrentPose()), f"scan {k}: pose bits of the shells differ from the pipeline's"
        assert int(cols[17]) == pipe.map.size(), f"scan {k}: map size"
        if k > 0:
            st = pipe.lastStats()
            assert [int(v) for v in cols[18:21]] == [st.valid, st.outlier, st.invalid], f"scan {k}: statistics pass"
    assert pipe.trackLoss() == 0 and pipe.getCurrentPose()[0, 3] > 3.0
