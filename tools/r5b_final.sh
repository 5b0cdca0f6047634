#!/bin/bash
# round 5, second session, closing run on the final tree (kernel sources unchanged since profile round r05): the GPU suite,
# bench.py as the driver runs it (default and --steps 20), and a 600-scan timed region for a tighter estimate of the rate
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > "$O/pytest_gpu.txt"; cat "$O/pytest_gpu.txt"
timeout 600 python bench.py 2>"$O/bench.err" | tail -1 > "$O/bench.json"
timeout 300 python bench.py --steps 20 2>"$O/bench20.err" | tail -1 > "$O/bench_driver_shape_steps20.json"
timeout 600 python bench.py --steps 600 --cpu-scans 0 --adapter-scans 0 --no-kernel-events 2>/dev/null | tail -1 > "$O/bench_600_steps.json"
python - <<PY
import json
for f in ("bench","bench_driver_shape_steps20","bench_600_steps"):
    d=json.load(open("$O/%s.json"%f)); r=d.get("roofline",{})
    print(f, round(d["value"],1), d["steps"], d.get("timed_call_us"), "roof", round(r.get("frac",0),4), r.get("traffic"), "host", round(d.get("host_vector_entry",{}).get("vs_resident",0),3), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
tail -14 "$O/bench.err"
