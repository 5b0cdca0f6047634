#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s8; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --cpu-scans 0 --adapter-scans 0 --no-kernel-events --no-host-vectors 2>/dev/null | tail -1 > "$O/b$i.json"
  python -c "
import json; d=json.load(open('$O/b$i.json')); print($i, round(d['value'],1), d['timed_call_us'], round(d['roofline']['avg_launch_us'],2))"
done
