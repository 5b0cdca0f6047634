"""The margin arithmetic of the visibility bits (k_update.hip, vis_may_pass) restated in numpy float32 and attacked with
random surfels and poses on the CPU: whenever phase 1a of the render passes (render_surfels.geom:76-103,
gen_indexmap.vert:62-81: front facing, range gate, field of view) lets a surfel through from a sensor pose inside the
margin, the conservative test must have kept it.  The kernel's copy of this arithmetic is what the `-m gpu` test
test_gpu_parity.py::test_visibility_lists_are_supersets checks on the device; this file checks the mathematics.
"""
import numpy as np

F = np.float32


def vis_may_pass(vertex, normal, ln, t0, r2, dt, dth, sin_lo, sin_hi, min_depth, max_depth):
    """numpy float32 twin of k_update.hip::vis_may_pass (same operations, same order)"""
    u = (vertex - t0).astype(F)
    r = np.sqrt((u * u).sum(1, dtype=F)).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_r = (F(1.0) / r).astype(F)
    alpha = (F(1.5708) * F(dt)) * inv_r
    dir_bounded = alpha < F(1.5708)
    f = -(normal * u).sum(1, dtype=F) * inv_r
    s = (u @ r2.astype(F)) * inv_r
    m = alpha + F(dth) + F(1e-3)
    rej = (f + alpha + F(1e-3)) * ln < F(0.009)
    rej = rej | (s < F(sin_lo) - m) | (s > F(sin_hi) + m)
    rej = rej & dir_bounded
    rej = rej | (r < F(min_depth) - F(dt)) | (r > F(max_depth) + F(dt))
    return ~rej


def rot(axis, angle):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def phase_1a(p, n, fov_up, fov_down, min_depth, max_depth):
    """what k_render lets through (float64: the margins under test are six orders above float32 rounding)"""
    depth = np.linalg.norm(p, axis=1)
    facing = (n * (-p / depth[:, None])).sum(1) > 0.01
    el_deg = np.degrees(np.arcsin(np.clip(p[:, 2] / depth, -1, 1)))
    in_fov = (el_deg >= -fov_down) & (el_deg < fov_up)
    in_range = (depth >= min_depth) & (depth <= max_depth)
    return facing & in_fov & in_range


def host_params(fov_up, fov_down, min_depth, max_depth):
    """vis_prepare of suma_api.hip"""
    lo, hi = -np.radians(fov_down), np.radians(fov_up)
    return dict(sin_lo=F(np.sin(lo)) - F(1e-4), sin_hi=F(np.sin(hi)) + F(1e-4), min_depth=F(min_depth) - F(0.01),
                max_depth=F(max_depth) + F(0.01))


def run_case(rng, n_pts, dt, dth, lim_dt, lim_th, inc_t, inc_r, near=False):
    fov_up, fov_down, min_depth, max_depth = 3.0, 25.0, 2.0, 75.0
    hp = host_params(fov_up, fov_down, min_depth, max_depth)
    # surfels in the frame of the update's pose: all ranges incl. very near the sensor and beyond the gate
    rr = rng.uniform(0.05, 3.0, n_pts) if near else np.exp(rng.uniform(np.log(0.2), np.log(120.0), n_pts))
    d = rng.normal(size=(n_pts, 3))
    d[:, 2] *= 0.35  # concentrate around the field of view
    d /= np.linalg.norm(d, axis=1)[:, None]
    v = d * rr[:, None]
    n = rng.normal(size=(n_pts, 3))
    n /= np.linalg.norm(n, axis=1)[:, None]
    ln = rng.uniform(0.999, 1.001, n_pts)  # k_render does not normalise the stored normal
    # predicted increment X0 and the actual pose X = X0 * D with D inside what k_render's pose test admits
    R0 = rot(rng.normal(size=3), rng.uniform(0, inc_r))
    t0 = rng.normal(size=3)
    t0 *= rng.uniform(0, inc_t) / np.linalg.norm(t0)
    kept = vis_may_pass(v.astype(F), n.astype(F), ln.astype(F), t0.astype(F), R0[:, 2].astype(F), dt, dth, **hp)
    worst = 0
    for _ in range(8):
        RD = rot(rng.normal(size=3), rng.uniform(0, lim_th))
        tD = rng.normal(size=3)
        tD *= (lim_dt * rng.uniform(0, 1) ** 0.2) / np.linalg.norm(tD)
        R, t = R0 @ RD, R0 @ tD + t0
        p = (v - t) @ R  # X^-1 v = R^T (v - t)
        nn = (n * ln[:, None]) @ R
        passes = phase_1a(p, nn, fov_up, fov_down, min_depth, max_depth)
        worst += int(np.count_nonzero(passes & ~kept))
    return worst, float(kept.mean()), float(passes.mean())


def test_next_pose_bit_is_a_superset():
    rng = np.random.default_rng(7)
    total = 0
    for k in range(40):
        bad, kept, passes = run_case(rng, 50_000, dt=0.45, dth=0.11, lim_dt=0.42, lim_th=0.105, inc_t=3.0, inc_r=0.3, near=(k % 5 == 4))
        total += bad
    assert total == 0


def test_update_pose_bit_is_a_superset():
    rng = np.random.default_rng(11)
    total = 0
    for k in range(40):
        bad, kept, passes = run_case(rng, 50_000, dt=0.02, dth=0.01, lim_dt=0.01, lim_th=0.004, inc_t=0.0, inc_r=0.0, near=(k % 5 == 4))
        total += bad
    assert total == 0


def test_the_bits_do_reject():
    """not vacuous: with the margins of the update-pose bit the kept fraction is close to what phase 1a lets through"""
    rng = np.random.default_rng(3)
    bad, kept, passes = run_case(rng, 200_000, dt=0.02, dth=0.01, lim_dt=0.01, lim_th=0.004, inc_t=0.0, inc_r=0.0)
    assert bad == 0 and passes > 0.05 and kept < passes + 0.05
    bad, kept1, passes = run_case(rng, 200_000, dt=0.45, dth=0.11, lim_dt=0.42, lim_th=0.105, inc_t=1.5, inc_r=0.05)
    assert bad == 0 and kept1 < 0.75


def test_degenerate_inputs_keep_the_surfel():
    """what cannot be proven out of view stays on the list: NaN in position, normal or normal length (every comparison of
    the test is written so that a NaN fails it); a surfel AT the sensor is out of the range gate for certain"""
    hp = host_params(3.0, 25.0, 2.0, 75.0)
    v = np.array([[np.nan, 1, 1], [10, 0, -1], [10, 0, -1], [0, 0, 0]], dtype=F)
    n = np.array([[0, 0, 1], [np.nan, 0, 0], [1, 0, 0], [0, 0, 1]], dtype=F)  # [2]: faces away
    ln = np.array([1, 1, np.nan, 1], dtype=F)
    z = np.zeros(3, F)
    with np.errstate(all="ignore"):
        kept = vis_may_pass(v, n, ln, z, np.array([0, 0, 1], F), 0.45, 0.11, **hp)
    assert kept[0] and kept[1] and kept[2] and not kept[3]
