#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s9; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
for i in 1 2; do
  timeout 600 python bench.py 2>"$O/bench_$i.err" | tail -1 > "$O/bench_$i.json"; cp gpurun_out/bench_kernels.json "$O/bench_kernels_$i.json"
  timeout 300 python bench.py --steps 20 2>"$O/driver_$i.err" | tail -1 > "$O/driver_$i.json"
done
python - <<'PY'
import json
for f in ("bench_1","driver_1","bench_2","driver_2"):
    d=json.load(open(f"gpurun_out/s9/{f}.json")); h=d["host_vector_entry"]
    print(f, round(d["value"],1), d["timed_call_us"], "roof", round(d["roofline"]["frac"],4), d["roofline"]["traffic"], "host", round(h["vs_resident"],3), h["call_us"]["max"], "cpu", round(d["cpu_baseline"]["value"],1), d["cpu_baseline"]["pose_bits_equal_gpu"], "adapter", d["adapter_path"]["classes_vs_phases"], d["adapter_path"]["resident_scans_per_s"])
PY
