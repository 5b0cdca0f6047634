#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s7; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 600 python -m pytest tests -m gpu -q -x -k "render or pipeline_process_scan or long_sequence_parity or gl_golden or index_above or update_variants" 2>&1 | tail -3
bash tools/ab.sh tools/libsuma_k4pf.bin semantic_suma_amd/libsuma_hip.so > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
bash tools/ab_kernels.sh tools/libsuma_k4pf.bin semantic_suma_amd/libsuma_hip.so > "$O/abk.txt" 2>&1; tail -13 "$O/abk.txt"
