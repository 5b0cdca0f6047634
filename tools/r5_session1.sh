#!/bin/bash
# round-5 GPU session 1: suite on the new build, the driver-shaped bench three times (host-vector call times), A/B of the
# exact barycentric division and of the combined arithmetic-specification build (timing only)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s1; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$O/pytest_gpu.txt"; cat "$O/pytest_gpu.txt" | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --cpu-scans 0 --adapter-scans 0 --no-kernel-events 2>/dev/null | tail -1 > "$O/bench20_$i.json"
  python - "$O/bench20_$i.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); h=d["host_vector_entry"]
print("run", sys.argv[1][-6], "value", round(d["value"],1), "host", round(h["value"],1), "ratio", round(h["vs_resident"],3), h["call_us"]["median"], h["call_us"]["max"], h["call_us"]["calls_over_twice_the_median"], h["call_us"]["first_30"][:12])
PY
done
bash tools/ab.sh tools/libsuma_plaindiv.bin semantic_suma_amd/libsuma_hip.so tools/libsuma_v2.bin > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
bash tools/ab_kernels.sh semantic_suma_amd/libsuma_hip.so tools/libsuma_v2.bin > "$O/abk_v2.txt" 2>&1; tail -14 "$O/abk_v2.txt"
timeout 300 python bench.py --steps 20 2>"$O/bench_driver.err" | tail -1 > "$O/bench_driver.json"; tail -16 "$O/bench_driver.err"
