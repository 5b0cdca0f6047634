"""Micro-benchmark: Frame2Model::jacobianProducts (one K6 pixel launch + the closing consume) on a 64x2048 frame pair."""
import sys, os, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import get_scan
from semantic_suma_amd import core
from semantic_suma_amd.types import params_with_size
W = 2048
p = params_with_size(W, max_iterations=10)
ctx = core.Context(p)
pre = core.Preprocessing(ctx)
f0, f1 = core.Frame(ctx, W, 64), core.Frame(ctx, W, 64)
pre.process(*get_scan(0, W)[:1], f0, *get_scan(0, W)[1:3], 20)
pre.process(*get_scan(1, W)[:1], f1, *get_scan(1, W)[1:3], 21)
obj = core.Frame2Model(ctx); obj.setData(f1, f0)
gn = core.LieGaussNewton(ctx)
T = np.eye(4); T[0, 3] = 1.0
obj.initialize(T)
for _ in range(3): obj.jacobianProducts()
ctx.profile(True); ctx.profile_reset()
t = time.perf_counter()
for _ in range(200): obj.jacobianProducts()
dt = time.perf_counter() - t
for k in ctx.profile_get(): print(k['name'], k['launches'], round(1000 * k['total_ms'] / max(k['launches'], 1), 2), 'us')
print('wall per eval call us', 1e6 * dt / 200, 'valid', obj.valid())
