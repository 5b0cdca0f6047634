"""Where the two surfel passes spend their time: builds libsuma_hip_timing.bin (-DSUMA_PHASE_TIMING: thread 0 of every
block accumulates wall_clock64 between the stations of its tiles / trips), runs the bench sequence until the map has
reached its steady size, then prints per kernel the average time one block spends in each phase of a launch (us).
Blocks run side by side, so the phases of one block add up to about the launch duration."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "semantic_suma_amd", "csrc")
lib = os.path.join(ROOT, "tools", "libsuma_hip_timing.bin")
srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and f != "suma_dist.hip"]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w",
                       "-DSUMA_PHASE_TIMING", "-shared", "-o", lib] + srcs + ["-lpthread"])
os.environ["SUMA_HIP_LIB"] = lib
from semantic_suma_amd import core, synth
from semantic_suma_amd.types import params_with_size
W = 2048
warm, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (110, 30)
pipe = core.SurfelMapping(params_with_size(W))
L = core.lib()
for f in (L.suma_debug_k4_phases, L.suma_debug_k9_phases):
    f.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((8192, 9), dtype=np.uint64)
for k in range(warm + n):
    if k == warm:
        pipe.ctx.synchronize()
        assert L.suma_debug_k4_phases(buf.ctypes.data, 1) == 0 and L.suma_debug_k9_phases(buf.ctypes.data, 1) == 0
    pts, lab, prob, _ = synth.generate_scan(k, n_azimuth=W)
    pipe.processScan(pts, lab, prob, fixed_iterations=10)
pipe.ctx.synchronize()
print(f"map: {pipe.map.size()} surfels; {n} scans measured")
names4 = ["surfel loads arrive", "1a transform/gate/project + rank barrier", "candidate list + barrier", "1b corners",
          "block prefix (2 barriers)", "2 pixel tests, z reads, atomics", "closing barrier", "loop / tail"]
names9 = ["ticket (atomic + 2 barriers)", "surfel loads + prepare", "pose entry, gather, update, LDS record", "ranks + publish",
          "look-back collect + barrier", "stream-out issue", "-", "loop / tail"]
for name, fn, names, launches in (("k_render (two slots per tile)", L.suma_debug_k4_phases, names4, 2 * n), ("k9_update", L.suma_debug_k9_phases, names9, n)):
    assert fn(buf.ctypes.data, 0) == 0
    blocks = int(buf[:, 8].sum())
    per_block = buf[:, :8].sum(axis=0).astype(np.float64) / max(1, blocks) / 100.0
    print(f"{name}: {blocks / launches:.0f} blocks per launch, sum of phases {per_block.sum():.1f} us per block and launch")
    for k in range(8):
        if names[k] != "-":
            print(f"  {names[k]:<44}{per_block[k]:7.2f} us  {100 * per_block[k] / per_block.sum():5.1f} %")
