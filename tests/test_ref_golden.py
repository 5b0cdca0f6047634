"""Golden vectors produced by the reference's OWN shaders (tests/golden/ref_180x16.npz and, at the bench geometry of
BASELINE configs[1], tests/golden/ref_2048x64.npz; made by tests/golden/make_ref_golden.py from oracle/_ref = the GLSL of /root/reference/src/shader compiled with g++; no oracle
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

HERE = os.path.dirname(os.path.abspath(__file__))


class Fixture:
    """one file of reference-shader vectors.  180x16: inputs stored in the file.  2048x64 (the bench geometry, BASELINE
    configs[1]): the inputs are the deterministic synthetic scans 0 and 2, regenerated here and checked against the
    checksum the generator stored -- so the GPU leg at the bench size does not pass through the oracle either."""

    def __init__(self, name, max_surfels):
        self.G = np.load(os.path.join(HERE, "golden", name))
        self.W, self.H = int(self.G["W"]), int(self.G["H"])
        self.P = params_with_size(self.W, self.H, max_surfels=max_surfels, max_poses=64)
        self._in = None

    def inputs(self, k):
        if "pts0" in self.G.files:
            return self.G[f"pts{k}"], self.G[f"lab{k}"], self.G[f"prob{k}"]
        if self._in is None:
            import hashlib
            from semantic_suma_amd import synth
            s0 = synth.generate_scan(0, n_azimuth=self.W, height=self.H)[:3]
            s1 = synth.generate_scan(2, n_azimuth=self.W, height=self.H)[:3]
            h = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (*s0, *s1))).digest()
            assert h == bytes(self.G["inputs_sha256"].tobytes()), "synthetic scans differ from the ones the fixture was made from"
            self._in = (s0, s1)
        return self._in[k]


FIXTURES = {"180x16": ("ref_180x16.npz", 1 << 16), "2048x64": ("ref_2048x64.npz", 1 << 19)}
_LOADED = {}


def fixture(name):
    if name not in _LOADED:
        _LOADED[name] = Fixture(*FIXTURES[name])
    return _LOADED[name]


W, H = 180, 16  # the filter fixture below


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


def check(fx, preprocess, update, k6):
    """preprocess(k, timestamp) -> (v, n, s); update(pose, k) -> (index_map, radius_conf, integrated, surfels);
    k6(T) -> acc words"""
    G = fx.G
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


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_oracle_reproduces_reference_shader_vectors(oracle_lib, name):
    fx = fixture(name)
    ora = oracle_lib.Oracle(fx.P, threads=4)
    frames = {}

    def preprocess(k, t):
        f = ora.preprocess(*fx.inputs(k), t, ora.frame())
        frames[k] = f
        return f.vertex, f.normal, f.semantic

    def update(pose, k):
        ora.map_update(pose, frames[k])
        return ora.map_index_map(), ora.map_radius_conf(), ora.map_integrated(), ora.map_surfels()

    def k6(T):
        return ora.jacobian_products(frames[1], frames[0], T, 0)[1]

    check(fx, preprocess, update, k6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_hip_reproduces_reference_shader_vectors(name):
    from semantic_suma_amd import core
    fx = fixture(name)
    ctx = core.Context(fx.P)
    pre, smap = core.Preprocessing(ctx), core.SurfelMap(ctx)
    frames = {}

    def preprocess(k, t):
        f = core.Frame(ctx, fx.W, fx.H)
        pts, lab, prob = fx.inputs(k)
        pre.process(pts, f, lab, prob, t)
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

    check(fx, preprocess, update, k6)


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
