#!/bin/bash
# round 5, second session, GPU call 1: the visibility lists -- their tests, the parity cases around them, an interleaved
# A/B of the scan rate with and without them, the per-kernel tables of both
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/b1; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "visibility or pipeline_process_scan or fallback or render_variants or submap_paging or update_variants or loop_closure or model_image_differs or capacity" 2>&1 | tail -15 > "$O/pytest_sel.txt"
cat "$O/pytest_sel.txt"
bash tools/ab_env.sh SUMA_NO_VIS_LISTS=1 > "$O/ab.txt" 2>&1; cat "$O/ab.txt"
B="python bench.py --cpu-scans 0 --adapter-scans 0 --no-host-vectors --profile-scans 40"
$B --kernels-json "$O/k_lists.json" 2>"$O/k_lists.err" | tail -1 > "$O/bench_lists.json"; tail -16 "$O/k_lists.err"
SUMA_NO_VIS_LISTS=1 $B --kernels-json "$O/k_nolists.json" 2>"$O/k_nolists.err" | tail -1 > "$O/bench_nolists.json"; tail -16 "$O/k_nolists.err"
python -c "
import json
for f in ('lists','nolists'):
    d=json.load(open('$O/bench_%s.json'%f)); print(f, round(d['value'],1), d['visibility_lists'], d['timed_call_us'])
"
