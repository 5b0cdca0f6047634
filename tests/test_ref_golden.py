"""Golden vectors produced by the reference's OWN shaders (tests/golden/ref_180x16.npz, made by
tests/golden/make_ref_golden.py from oracle/_ref = the GLSL of /root/reference/src/shader compiled with g++; no oracle
code involved).  CPU: the oracle reproduces them.  GPU (-m gpu): the HIP path reproduces them through the C-ABI --
a comparison of the product with the reference's shader arithmetic that does not pass through the oracle.
Equal VALUES are demanded on every field (a shader transforms directions with a w = 0 column that adds a signed
zero); records whose normal the GLSL slerp turned into NaN are exempt in nx / ny / nz (documented deviation; none
occur in this fixture).

tests/golden/ref_filters_180x16.npz holds the same for Preprocessing::process with the optional vertex-map filters on
(blended K1 + avg_vertexmap.frag, bilateral_filter.frag; both sampling states of `filter_sampling`)."""
import os

import numpy as np
import pytest

from semantic_suma_amd.types import params_with_size

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_180x16.npz"))
W, H = int(G["W"]), int(G["H"])
P = params_with_size(W, H, max_surfels=1 << 16, max_poses=64)


GF = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_filters_180x16.npz"))
FILTER_CASES = {
    "avg": dict(avg_vertexmap=1),
    "avg_nearest": dict(avg_vertexmap=1, filter_sampling=1),
    "bilateral": dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5, bilateral_sigma_range=2.5),
    "bilateral_nearest": dict(filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=4.5,
                              bilateral_sigma_range=2.5, filter_sampling=1),
    "avg_bilateral_nearest": dict(avg_vertexmap=1, filter_vertexmap=1, use_filtered_vertexmap=1, bilateral_sigma_space=2.0,
                                  bilateral_sigma_range=0.5, filter_sampling=1),
}  # as tests/golden/make_ref_golden.py


def eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ne = a != b
    if a.dtype.kind == "f":
        ne &= ~(np.isnan(a) & np.isnan(b))
    assert not ne.any(), f"{what}: {int(ne.sum())} of {ne.size} values differ, first at {np.argwhere(ne)[:3].tolist()}"


def eq_map(got, want, nan, what):
    assert got.shape == want.shape, f"{what}: {got.shape[0]} vs {want.shape[0]} surfels"
    for name in want.dtype.names:
        a, b = got[name].copy(), want[name].copy()
        if name in ("nx", "ny", "nz"):
            a[nan] = 0
            b[nan] = 0
        eq(a, b, f"{what}.{name}")


def check(preprocess, update, k6):
    """preprocess(k, timestamp) -> (v, n, s); update(pose, k) -> (index_map, radius_conf, integrated, surfels);
    k6(T) -> acc words"""
    for k in (0, 1):
        v, n, s = preprocess(k, k)
        eq(v, G[f"vertex{k}"], f"K1 vertex map {k}")
        eq(n, G[f"normal{k}"], f"K2 normal map {k}")
        eq(s, G[f"semantic{k}"], f"K3 semantic map {k}")
    _, _, _, m0 = update(np.eye(4), 0)
    eq_map(m0, G["map0"], np.zeros(m0.shape[0], bool), "map after the first update")
    idx, rc, mask, m1 = update(G["P1"], 1)
    eq(idx, G["idx1"], "K7 index map")
    eq(rc, G["rc1"], "K8 radius map")
    eq(mask != 0, G["mask1"] != 0, "K9 integration mask")
    eq_map(m1, G["map1"], G["nan1"], "map after the second update")
    eq(k6(G["k6_T"]), G["k6_acc"], "K6 accumulator words")
    assert G["map1"].shape[0] > 1500 and int(G["mask1"].sum()) > 200 and int(G["k6_acc"][29]) > 300


def test_oracle_reproduces_reference_shader_vectors(oracle_lib):
    ora = oracle_lib.Oracle(P)
    frames = {}

    def preprocess(k, t):
        f = ora.preprocess(G[f"pts{k}"], G[f"lab{k}"], G[f"prob{k}"], t, ora.frame())
        frames[k] = f
        return f.vertex, f.normal, f.semantic

    def update(pose, k):
        ora.map_update(pose, frames[k])
        return ora.map_index_map(), ora.map_radius_conf(), ora.map_integrated(), ora.map_surfels()

    def k6(T):
        return ora.jacobian_products(frames[1], frames[0], T, 0)[1]

    check(preprocess, update, k6)


@pytest.mark.gpu
def test_hip_reproduces_reference_shader_vectors():
    from semantic_suma_amd import core
    ctx = core.Context(P)
    pre, smap = core.Preprocessing(ctx), core.SurfelMap(ctx)
    frames = {}

    def preprocess(k, t):
        f = core.Frame(ctx, W, H)
        pre.process(G[f"pts{k}"], f, G[f"lab{k}"], G[f"prob{k}"], t)
        frames[k] = f
        return f.download(0), f.download(1), f.download(2)

    def update(pose, k):
        smap.update(pose, frames[k])
        return smap.index_map(), smap.radius_conf(), smap.integrated(), smap.getAllSurfels()

    def k6(T):
        obj = core.Frame2Model(ctx)
        obj.setData(frames[1], frames[0])
        obj.initialize(T)
        obj.jacobianProducts()
        return obj.acc

    check(preprocess, update, k6)


def check_filters(name, preprocess):
    """preprocess(timestamp) -> (v, n, s) of the dense scan with FILTER_CASES[name] set"""
    for t in (0, 12):
        v, n, s = preprocess(t)
        eq(v, GF[f"{name}_t{t}_vertex"], f"{name} t={t}: vertex map")
        eq(n, GF[f"{name}_t{t}_normal"], f"{name} t={t}: normal map")
        eq(s, GF[f"{name}_t{t}_semantic"], f"{name} t={t}: semantic map")
    v = GF[f"{name}_t12_vertex"]
    assert int((v[..., 3] > 0.5).sum()) > 0.8 * W * H and GF["pts"].shape[0] > 4 * W * H  # dense: sums of several points


@pytest.mark.parametrize("name", sorted(FILTER_CASES))
def test_oracle_reproduces_reference_filter_vectors(oracle_lib, name):
    ora = oracle_lib.Oracle(params_with_size(W, H, max_surfels=1 << 16, max_poses=64, **FILTER_CASES[name]), threads=4)

    def preprocess(t):
        f = ora.preprocess(GF["pts"], GF["lab"], GF["prob"], t, ora.frame())
        return f.vertex.copy(), f.normal.copy(), f.semantic.copy()  # the views die with the frame

    check_filters(name, preprocess)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FILTER_CASES))
def test_hip_reproduces_reference_filter_vectors(name):
    from semantic_suma_amd import core
    ctx = core.Context(params_with_size(W, H, max_surfels=1 << 16, max_poses=64, **FILTER_CASES[name]))
    pre = core.Preprocessing(ctx)

    def preprocess(t):
        f = core.Frame(ctx, W, H)
        pre.process(GF["pts"], f, GF["lab"], GF["prob"], t)
        return f.download(0), f.download(1), f.download(2)

    check_filters(name, preprocess)
