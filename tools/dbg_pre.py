import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import get_scan, bits
from semantic_suma_amd import core
from semantic_suma_amd.types import params_with_size
from oracle import pyoracle
p = params_with_size(900)
pts, lab, prob, _ = get_scan(0, 900, False)
ctx = core.Context(p); ora = pyoracle.Oracle(p)
hf = core.Frame(ctx, 900, 64)
core.Preprocessing(ctx).process(pts, hf, lab, prob, 0)
of = ora.preprocess(pts, lab, prob, 0, ora.frame())
a = hf.download(1); b = of.map(1)
ne = bits(a) != bits(b)
idx = np.argwhere(ne.any(axis=2))
print(len(idx), 'pixels differ')
nan_both = np.isnan(a) & np.isnan(b)
print('nan-both among differing:', int((ne & nan_both).sum()), 'of', int(ne.sum()))
for (y, x) in idx[:8]:
    print(y, x, a[y, x], b[y, x], [hex(v) for v in bits(a[y, x])], [hex(v) for v in bits(b[y, x])])
d = np.abs(a - b)[ne & ~nan_both]
print('max abs diff non-nan', d.max() if d.size else 0)
