#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s5; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 300 python tools/host_entry_idle_probe.py 2>&1 | tail -1 > "$O/host_entry_idle_probe.json"; cat "$O/host_entry_idle_probe.json"; echo
timeout 300 python -m pytest tests -m gpu -q -k "native_scan_loop or frame_behind or history_belongs" 2>&1 | tail -3
timeout 600 python bench.py 2>"$O/bench.err" | tail -1 > "$O/bench.json"; cp gpurun_out/bench_kernels.json "$O/bench_kernels_hip_events.json"; tail -15 "$O/bench.err"
for i in 1 2; do timeout 300 python bench.py --steps 20 2>"$O/bench_driver_$i.err" | tail -1 > "$O/bench_driver_$i.json"; done
python - <<'PY'
import json
for f in ("bench","bench_driver_1","bench_driver_2"):
    d=json.load(open(f"gpurun_out/s5/{f}.json")); h=d["host_vector_entry"]
    print(f, round(d["value"],1), "roof", round(d["roofline"]["frac"],4), d["roofline"]["traffic"], "host", round(h["vs_resident"],3), h["call_us"]["max"], h["call_us"]["untimed_warm_up_calls"], "adapter", d.get("adapter_path",{}).get("classes_vs_phases"), d.get("adapter_path",{}).get("resident_scans_per_s"))
PY
