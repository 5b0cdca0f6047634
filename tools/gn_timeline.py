"""Intra-launch timeline of k_icp_step: builds libsuma_hip_timing.so (-DSUMA_GN_TIMING: every block stamps
wall_clock64 at seven stations), runs one Gauss-Newton chain on a 64x2048 frame pair and prints, for the LAST pixel
launch, the median / max over blocks of each station relative to the earliest block start (us)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
csrc = os.path.join(ROOT, "semantic_suma_amd", "csrc")
lib = os.path.join(ROOT, "tools", "libsuma_hip_timing.bin")
srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and f != "suma_dist.hip"]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w",
                       "-DSUMA_GN_TIMING", "-shared", "-o", lib] + srcs + ["-lpthread"])
os.environ["SUMA_HIP_LIB"] = lib
from conftest import get_scan
from semantic_suma_amd import core
from semantic_suma_amd.types import params_with_size
W = 2048
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
p = params_with_size(W, max_iterations=n_iter, stopping_threshold=0.0, delta=0.0)
ctx = core.Context(p)
pre = core.Preprocessing(ctx)
f0, f1 = core.Frame(ctx, W, 64), core.Frame(ctx, W, 64)
s0, s1 = get_scan(0, W), get_scan(1, W)
pre.process(s0[0], f0, s0[1], s0[2], 20); pre.process(s1[0], f1, s1[1], s1[2], 21)
obj = core.Frame2Model(ctx); obj.setData(f1, f0)
gn = core.LieGaussNewton(ctx)
T0 = np.eye(4); T0[0, 3] = 1.0
L = core.lib()
L.suma_debug_gn_timing.argtypes = [C.c_void_p]
names = ["start", "loads arrived", "folded", "solved", "pose ready", "pixel+reduce", "adds issued", "exp formed"]
acc = []
for rep in range(6):
    gn.minimize(obj, T0, history_cap=0)
    t = np.zeros((256, 8), dtype=np.uint64)
    assert L.suma_debug_gn_timing(t.ctypes.data) == 0
    t = t[:, :8].astype(np.int64)
    if rep: acc.append((t - t[:, 0].min()) / 100.0)  # 100 MHz -> us
a = np.stack(acc)  # reps x blocks x stations
print(f"last pixel launch of a {n_iter}-iteration chain, us after the earliest block start (median over blocks / max over blocks, mean of {a.shape[0]} chains)")
for k, n in enumerate(names):
    print(f"  {n:<14}{np.median(a[:, :, k], axis=1).mean():7.2f} /{a[:, :, k].max(axis=1).mean():7.2f}")
print("  block 0:      " + "  ".join(f"{a[:, 0, k].mean():.2f}" for k in range(8)))
