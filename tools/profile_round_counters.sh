#!/bin/bash
# the legs of tools/profile_round.sh that bench.py's loop-closure leg would distort (its batched / converged launches are
# k_icp_step launches too): rocprofv3 kernel stats and the five PMC passes on `bench.py --no-loop-closure`, then -- with
# the refreshed profiles/hbm_traffic.json in place -- the default bench line, so that `roofline.traffic` is in it.
#   gpurun --timeout 2400 -- 'bash tools/profile_round_counters.sh r06'   then   bash tools/collect_profiles.sh r06
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 400 rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench --output-format csv -- python bench.py --cpu-scans 0 --adapter-scans 0 --no-loop-closure --no-reference-mode 2>/dev/null | tail -1 > "$O/bench_under_rocprof.json"
bash tools/pmc_refresh.sh "$TAG" > "$O/pmc_refresh.log" 2>&1
f1() { find "$O/$1" -name "$2" | head -1; }
python tools/rocprof_summary.py "$(f1 prof '*kernel_trace.csv')" > "$O/kernel_trace_summary.txt" 2>&1
cp "$(f1 prof '*kernel_stats.csv')" "$O/rocprofv3_kernel_stats.csv"
python tools/trace_gaps.py "$(f1 prof '*kernel_trace.csv')" 340 360 > "$O/scan_timeline_gaps.txt" 2>&1
python tools/make_hbm_traffic.py "$(f1 pmc_fetch '*counter_collection.csv')" "$(f1 pmc_write '*counter_collection.csv')" 2048 64 "$O/hbm_traffic.json" > "$O/hbm_traffic_pmc.txt" 2>&1
python tools/sq_summary.py "$(f1 pmc_sq1 '*counter_collection.csv')" "$(f1 pmc_sq2 '*counter_collection.csv')" "$(f1 pmc_sq3 '*counter_collection.csv')" > "$O/sq_summary.txt" 2>&1
rm -rf "$O/prof" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3"
cp "$O/hbm_traffic.json" profiles/hbm_traffic.json
timeout 600 python bench.py 2>"$O/bench.err" | tail -1 > "$O/bench.json"; cp gpurun_out/bench_kernels.json "$O/bench_kernels_hip_events.json"
timeout 300 python bench.py --steps 20 2>"$O/bench_driver_shape.err" | tail -1 > "$O/bench_driver_shape_steps20.json"
head -5 "$O/kernel_trace_summary.txt"; grep -E "k_icp_step" "$O/hbm_traffic_pmc.txt"; cut -c1-200 "$O/bench.json"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['roofline'])"
