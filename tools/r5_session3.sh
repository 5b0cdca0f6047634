#!/bin/bash
# round-5 GPU session 3: the new GPU tests, the bandwidth regime (50 M surfels) and the literal full sequence on the new build
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s3; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 600 python -m pytest tests -m gpu -q -s -k "acceptance_line or host_vector_entry or convergence_and_batch" 2>&1 | tail -12 > "$O/pytest_new.txt"; cat "$O/pytest_new.txt"
timeout 400 python tools/stress_map.py 2>&1 | tail -1 > "$O/stress.json"; python -c "
import json; d=json.load(open('$O/stress.json'))
for k,v in d['kernels'].items(): print(k, round(v['avg_ms'],3), 'ms', round(v['gbps']), 'GB/s', round(v['frac'],3))"
timeout 900 python bench.py --cpu-scans 0 --no-kernel-events --adapter-scans 0 --no-host-vectors --steps 4541 --warmup 0 --preroll 0 --max-surfels 16777216 2>/dev/null | tail -1 > "$O/bench_full_sequence_4541.json"; cut -c1-300 "$O/bench_full_sequence_4541.json"
