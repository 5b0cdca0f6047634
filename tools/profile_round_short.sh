#!/bin/bash
# The part of tools/profile_round.sh that depends on the render kernel (round 5, second session: k_render's edge functions
# moved to exact binary64 with 20 GPU-minutes left in the round): GPU suite, bench (default / driver shape / literal full
# sequence), rocprofv3 kernel stats, the HBM and SQ counter passes, the 50 M stress run, the phase timeline, and LAST the
# 4541-scan parity check.  The modes that do not touch the map passes' evidence (hypotheses, sequences11, adapter-300,
# ingest, multi-pipeline, gloo two-rank, Gauss-Newton timeline) keep the files of the full round taken earlier.
#   gpurun --timeout 1700 -- 'bash tools/profile_round_short.sh r05'   then   bash tools/collect_profiles.sh r05
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
B="python bench.py --cpu-scans 0 --no-kernel-events --adapter-scans 0 --no-loop-closure --no-reference-mode"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2 > "$O/pytest_gpu.txt"
timeout 600 python bench.py 2>"$O/bench.err" | tail -1 > "$O/bench.json"; cp gpurun_out/bench_kernels.json "$O/bench_kernels_hip_events.json"
timeout 300 python bench.py --steps 20 2>"$O/bench_driver_shape.err" | tail -1 > "$O/bench_driver_shape_steps20.json"
timeout 900 $B --no-host-vectors --steps 4541 --warmup 0 --preroll 0 --max-surfels 16777216 2>/dev/null | tail -1 > "$O/bench_full_sequence_4541.json"
timeout 400 rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench --output-format csv -- python bench.py --cpu-scans 0 --adapter-scans 0 --no-loop-closure --no-reference-mode 2>/dev/null | tail -1 > "$O/bench_under_rocprof.json"
bash tools/pmc_refresh.sh "$TAG" > "$O/pmc_refresh.log" 2>&1
timeout 400 python tools/stress_map.py 2>&1 | tail -1 > "$O/stress.json"
timeout 300 python tools/phase_timeline.py 250 30 2>&1 | tail -19 > "$O/phase_timeline.txt"
timeout 900 python tools/long_parity.py --check tests/golden/long_trace_4541.npz --out "$O/long_parity_4541_scans.json" 2>"$O/long_parity.err" | tail -1 | cut -c1-300
cat "$O/pytest_gpu.txt"; cut -c1-400 "$O/bench.json"; cut -c1-300 "$O/bench_full_sequence_4541.json"; tail -5 "$O/pmc_refresh.log"; ls "$O"
