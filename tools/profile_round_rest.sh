#!/bin/bash
# the lines of tools/profile_round.sh that tools/profile_round_short.sh leaves out (modes that do not depend on the render
# kernel), so that a short round + this = a full round:   gpurun --timeout 1500 -- 'bash tools/profile_round_rest.sh r05'
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 300 python bench.py --mode hypotheses --steps 60 2>/dev/null | tail -1 > "$O/bench_hypotheses.json"
SUMA_SEQ_CONCURRENT=2 timeout 400 python bench.py --mode sequences11 2>/dev/null | tail -1 > "$O/bench_sequences11.json"
timeout 300 python bench.py --mode adapter --adapter-scans 300 2>/dev/null | tail -1 > "$O/adapter_path_300_scans.json"
timeout 300 python tools/ingest_bench.py 2>/dev/null | tail -1 > "$O/ingest.json"
timeout 300 python tools/multi_seq.py 4 60 2>/dev/null | tail -1 > "$O/multi_seq.txt"
SUMA_BENCH_FORCE_DEVICE=0 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --cpu-scans 0 --adapter-scans 0 --no-kernel-events 2>/dev/null | tail -1 > "$O/bench_gpus2_self_launched_gloo.json"
timeout 300 python tools/gn_timeline.py 2>&1 | tail -14 > "$O/gn_timeline.txt"
for f in bench_hypotheses bench_sequences11 adapter_path_300_scans ingest bench_gpus2_self_launched_gloo; do echo "$f: $(cut -c1-160 $O/$f.json)"; done; cat "$O/multi_seq.txt" | cut -c1-200; tail -12 "$O/gn_timeline.txt"
