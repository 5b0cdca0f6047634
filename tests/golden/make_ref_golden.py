#!/usr/bin/env python
"""Generates tests/golden/ref_180x16.npz FROM THE REFERENCE'S OWN SHADERS (oracle/_ref: the GLSL of
/root/reference/src/shader compiled with g++ by oracle/ref_build.py; oracle/pyref.py sets the uniforms as the
reference's host code does).  No oracle code is involved in producing these vectors:

  frame0 / frame1   Preprocessing::process of two synthetic scans (gen_vertexmap, gen_normalmap, floodfill)
  map0              SurfelMap::update(identity, frame0) on the empty map: K7, K8, K9, K10, K11 chained through the shaders
  idx1, rc1, mask1, map1
                    SurfelMap::update(P1, frame1) on map0 with the ground-truth pose P1: index map, radius map,
                    integration mask, resulting map; nan1 flags records whose normal the GLSL slerp turned into NaN
                    (update_surfels.vert:113-124; the one documented behavioural deviation)
  k6_acc            Frame2Model_jacobians at pose k6_T, data = frame1, model = frame0, one entry per invocation, every
                    emitted term summed in 2^-28 fixed point
and tests/golden/ref_filters_180x16.npz: Preprocessing::process with the optional vertex-map filters
(Preprocessing.cpp:150-236: blended K1 + avg_vertexmap.frag, bilateral_filter.frag) on a DENSE scan (about six points
per texel, so the blended sums have several terms), one case per FILTER_CASES entry: vertex, normal, semantic map.
The GPU suite compares the HIP path with these vectors directly (tests/test_ref_golden.py); /root/reference is only
needed here, at generation time.  Run from the repo root:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyref  # noqa: E402
from semantic_suma_amd import synth  # noqa: E402
from semantic_suma_amd.types import SURFEL_DTYPE, params_with_size  # noqa: E402
from test_ref_shaders import unpack_fix  # noqa: E402

W, H = 180, 16


FILTER_CASES = {
    "avg": dict(avg_vertexmap=1),
    "avg_nearest": dict(avg_vertexmap=1, filter_sampling=1),
    "bilateral": dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5, bilateral_sigma_range=2.5),
    "bilateral_nearest": dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5,
                              bilateral_sigma_range=2.5, filter_sampling=1),
    "avg_bilateral_nearest": dict(avg_vertexmap=1, filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=2.0,
                                  bilateral_sigma_range=0.5, filter_sampling=1),
}


def filters():
    pts, lab, prob, _ = synth.generate_scan(3, n_azimuth=3 * W, height=2 * H)
    out = {"W": W, "H": H, "pts": pts, "lab": lab, "prob": prob}
    for name, ov in FILTER_CASES.items():
        p = params_with_size(W, H, max_surfels=1 << 16, max_poses=64, **ov)
        ref = pyref.Ref(p)
        for t in (0, 12):  # isfirst on / off (Preprocessing.cpp:176-179)
            v, n, s = ref.preprocess(pts, lab, prob, t)
            out[f"{name}_t{t}_vertex"], out[f"{name}_t{t}_normal"], out[f"{name}_t{t}_semantic"] = v, n, s
    path = os.path.join(ROOT, "tests", "golden", "ref_filters_180x16.npz")
    np.savez_compressed(path, **out)
    v = out["avg_t12_vertex"]
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KB; {pts.shape[0]} points, {int((v[..., 3] > 0.5).sum())} valid texels")


def main():
    p = params_with_size(W, H, max_surfels=1 << 16, max_poses=64)
    ref = pyref.Ref(p)
    s0 = synth.generate_scan(0, n_azimuth=W, height=H)
    s1 = synth.generate_scan(2, n_azimuth=W, height=H)  # two steps ahead: a visible motion
    P1 = (np.linalg.inv(s0[3]) @ s1[3]).astype(np.float32)
    out = {"W": W, "H": H, "pts0": s0[0], "lab0": s0[1], "prob0": s0[2], "pts1": s1[0], "lab1": s1[1], "prob1": s1[2],
           "P1": P1}
    f0 = ref.preprocess(s0[0], s0[1], s0[2], 0)
    f1 = ref.preprocess(s1[0], s1[1], s1[2], 1)
    for k, f in ((0, f0), (1, f1)):
        out[f"vertex{k}"], out[f"normal{k}"], out[f"semantic{k}"] = f
    poses = np.tile(np.eye(4, dtype=np.float32).T.reshape(16), (p.max_poses, 1))  # column-major identity table

    def update(surfels, pose, t, frame):
        poses[t] = np.asarray(pose, dtype=np.float32).T.reshape(16)  # poses_[timestamp_] = pose, SurfelMap.cpp:494
        idx = ref.indexmap(surfels, poses, pose)
        rc = ref.radius_conf(frame[0], frame[1])
        upd, mask = ref.update(surfels, poses, pose, t, frame, rc, idx)
        gen = ref.generate(frame, rc, mask, pose, t)
        ext = np.float32(2.0) * np.float32(p.submap_dimension) * np.float32(p.submap_extent) + np.float32(p.submap_extent)
        new = ref.copy(upd, gen, poses, (np.float32(0), np.float32(0)), ext)
        return idx, rc, mask, upd, new

    empty = np.zeros(0, dtype=SURFEL_DTYPE)
    _, _, _, _, map0 = update(empty, np.eye(4, dtype=np.float32), 0, f0)
    idx1, rc1, mask1, upd1, map1 = update(map0, P1, 1, f1)
    nan1 = np.isnan(map1["nx"])
    out.update(map0=map0, idx1=idx1.astype(np.uint32), rc1=rc1, mask1=(mask1[:, :, 0] > 0.5).astype(np.uint8), map1=map1,
               nan1=nan1, n_updated1=np.uint32(upd1.shape[0]))
    T = np.eye(4)
    T[:3, 3] = [2.1, 0.03, 0.0]
    _, fix = ref.jacobians(f1, f0, T, 0, entries_per_kernel=1)
    out["k6_T"] = T
    out["k6_acc"] = unpack_fix(fix)[0]
    path = os.path.join(ROOT, "tests", "golden", "ref_180x16.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KB; map0 {map0.shape[0]} surfels, map1 {map1.shape[0]} "
          f"({int(nan1.sum())} with NaN normal), {int(out['mask1'].sum())} pixels integrated, K6 valid {int(out['k6_acc'][29])}")


if __name__ == "__main__":
    main()
    filters()
