#!/bin/bash
# the host-vector entry (the reference's processScan(const rv::Laserscan&)) against the CPUs the
# host grants -- bench.py's host_vector_entry under taskset with 16 / 8 / 4 / 2 CPUs.  Writes one JSON line per setting.
out=${1:-gpurun_out/r06_host_entry_cpus.jsonl}
: > "$out"
for cpus in 0-15 0-7 0-3 0-1; do
  taskset -c $cpus python bench.py --cpu-scans 0 --adapter-scans 0 --no-kernel-events --no-loop-closure 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
h = d['host_vector_entry']
print(json.dumps({'taskset': '$cpus', 'resident_scans_per_s': round(d['value'], 1), 'host_vector_scans_per_s': round(h['value'], 1),
                  'vs_resident': round(h['vs_resident'], 4), 'call_us': {k: h['call_us'][k] for k in ('median', 'max')},
                  'call_breakdown_us': h['call_breakdown_us'], 'cpus': h['cpus']}))" | tee -a "$out"
done
