"""BASELINE config 3 on one GPU: n_hyp Gauss-Newton chains of the same 64x2048 frame pair in one batch
(gridDim.y = hypothesis).  Reports time per chain launch and the algorithmic bandwidth of k_icp_step."""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import get_scan
from semantic_suma_amd import core
from semantic_suma_amd.distributed import hypothesis_starts
from semantic_suma_amd.types import params_with_size
W = 2048
p = params_with_size(W, max_iterations=10, stopping_threshold=0.0, delta=0.0)
ctx = core.Context(p)
pre = core.Preprocessing(ctx)
f0, f1 = core.Frame(ctx, W, 64), core.Frame(ctx, W, 64)
s0, s1 = get_scan(0, W), get_scan(1, W)
pre.process(s0[0], f0, s0[1], s0[2], 20)
pre.process(s1[0], f1, s1[1], s1[2], 21)
obj = core.Frame2Model(ctx); obj.setData(f1, f0)
gn = core.LieGaussNewton(ctx)
T0 = np.eye(4); T0[0, 3] = 1.0
for n in (1, 2, 4, 8, 16, 32):
    starts = hypothesis_starts(T0, n)
    gn.minimize_batch(starts, obj)
    ctx.profile(1); ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(10): Ts, st = gn.minimize_batch(starts, obj)
    dt = (time.perf_counter() - t) / 10
    k = [q for q in ctx.profile_get() if q['name'] == 'k6_icp_step'][0]
    us = 1000 * k['total_ms'] / k['launches']
    print(f"n_hyp={n:2d}: {us:7.2f} us per GN launch, {k['bytes'] / k['total_ms'] / 1e6:7.0f} GB/s algorithmic ({k['bytes'] / k['total_ms'] / 8e9:.3f} of 8 TB/s), {1e6 * dt:8.0f} us per batch of chains, {n / dt:7.0f} hypotheses/s")
    ctx.profile(0)
