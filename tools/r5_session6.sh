#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s6; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
SUMA_HIP_LIB=tools/libsuma_k4pf.bin timeout 600 python -m pytest tests -m gpu -q -x -k "render or pipeline_process_scan or long_sequence_parity or gl_golden or index_above" 2>&1 | tail -3
bash tools/ab.sh semantic_suma_amd/libsuma_hip.so tools/libsuma_k4pf.bin > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
bash tools/ab_kernels.sh semantic_suma_amd/libsuma_hip.so tools/libsuma_k4pf.bin > "$O/abk.txt" 2>&1; tail -13 "$O/abk.txt"
for L in semantic_suma_amd/libsuma_hip.so tools/libsuma_k4pf.bin; do SUMA_HIP_LIB=$L timeout 400 python tools/stress_map.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('$L', {k:(round(v['avg_ms'],3), round(v['frac'],3)) for k,v in d['kernels'].items() if k.startswith(('k4','k9','k7'))})"; done
