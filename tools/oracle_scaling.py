"""CPU oracle throughput vs OpenMP thread count (same 64x2048 sequence as bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from semantic_suma_amd import synth
from semantic_suma_amd.types import params_with_size
W = 2048
p = params_with_size(W)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scans = [synth.generate_scan(k, n_azimuth=W) for k in range(n)]
for th in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,8,16,32,64,128".split(","))]:
    op = pyoracle.OraclePipeline(p, threads=th)
    t = time.perf_counter()
    for pts, lab, prob, _ in scans:
        op.process_scan(pts, lab, prob, fixed_iterations=10)
    print(f"{th:4d} threads: {n / (time.perf_counter() - t):6.1f} scans/s", flush=True)
