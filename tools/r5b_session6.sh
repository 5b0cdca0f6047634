#!/bin/bash
# round 5, second session: look-ahead preprocessing in the native loop -- its tests, the 4541-scan trace through both
# entries, an interleaved A/B (SUMA_NO_LOOKAHEAD=1 = the previous schedule), per-kernel tables
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/b6; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long.py -m gpu -x -q -k "lookahead or native_scan_loop or fallback or pipeline_process_scan or full_sequence or host_vector" 2>&1 | tail -6 > "$O/pytest_sel.txt"
cat "$O/pytest_sel.txt"
bash tools/ab_env.sh SUMA_NO_LOOKAHEAD=1 > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
B="python bench.py --cpu-scans 0 --adapter-scans 0 --no-host-vectors --profile-scans 40"
$B 2>"$O/k_ahead.err" | tail -1 > "$O/bench_ahead.json"; tail -14 "$O/k_ahead.err"
