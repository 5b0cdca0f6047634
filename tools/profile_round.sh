#!/bin/bash
# One-call evidence run for profiles/ (inside a gpurun call, ~40 s of GPU time):
#   gpurun --timeout 1800 -- 'bash tools/profile_round.sh r01d'
# writes gpurun_out/<tag>/: pytest, bench (60 and 300 steps), rocprofv3 kernel trace + stats, the two PMC passes
# (counter runs WITHOUT any trace domain, as the pool requires) and the 50 M-surfel stress run.
# Afterwards, on the build machine: copy into profiles/ with the r0N_ prefix (see profiles/README.md) and run
#   python tools/rocprof_summary.py <tag>/prof/bench_kernel_trace.csv ; python tools/make_hbm_traffic.py <fetch> <write> 2048 64 profiles/hbm_traffic.json
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p "$O"
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -2 > "$O/pytest_gpu.txt"
timeout 300 python bench.py 2>/dev/null | tail -1 > "$O/bench.json"; cp gpurun_out/bench_kernels.json "$O/bench_kernels_hip_events.json"
timeout 300 python bench.py --steps 300 --cpu-scans 0 --no-kernel-events 2>/dev/null | tail -1 > "$O/bench_300_steps.json"
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench --output-format csv -- python bench.py --cpu-scans 0 2>/dev/null | tail -1 > "$O/bench_under_rocprof.json"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$O/pmc_fetch" -o f --output-format csv -- python bench.py --cpu-scans 0 --no-kernel-events --steps 30 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$O/pmc_write" -o w --output-format csv -- python bench.py --cpu-scans 0 --no-kernel-events --steps 30 > /dev/null 2>&1
timeout 400 python tools/stress_map.py 2>&1 | tail -1 > "$O/stress.json"
cat "$O/pytest_gpu.txt"; cut -c1-200 "$O/bench.json"
