#!/bin/bash
# One-call evidence run for profiles/ (inside a gpurun call, ~12 min of GPU time):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r03'
# writes gpurun_out/<tag>/: pytest (incl. the 50 M-surfel case of BASELINE configs[4]), bench (default = timed at the
# steady map after a 300-scan pre-roll, with the adapter-path and CPU legs), the literal full sequence (--steps 4541),
# rocprofv3 kernel trace + stats, the two HBM PMC passes and three SQ-counter passes at the steady state (counter runs
# WITHOUT any trace domain, as the pool requires), the 50 M-surfel stress run, the config-3 / config-4 modes, the
# host-scan hand-over and the multi-pipeline run.  Afterwards, on the build machine:  bash tools/collect_profiles.sh <tag>
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
B="python bench.py --cpu-scans 0 --no-kernel-events --adapter-scans 0 --no-loop-closure --no-reference-mode"
# the driver's exact command first (round-5 review item 3): -x, so that a failure anywhere shows as what the driver would record
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > "$O/pytest_gpu.txt"
timeout 600 python bench.py 2>"$O/bench.err" | tail -1 > "$O/bench.json"; cp gpurun_out/bench_kernels.json "$O/bench_kernels_hip_events.json"
timeout 300 python bench.py --steps 20 2>"$O/bench_driver_shape.err" | tail -1 > "$O/bench_driver_shape_steps20.json"
timeout 900 $B --no-host-vectors --steps 4541 --warmup 0 --preroll 0 --max-surfels 16777216 2>/dev/null | tail -1 > "$O/bench_full_sequence_4541.json"
timeout 400 rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench --output-format csv -- python bench.py --cpu-scans 0 --adapter-scans 0 --no-loop-closure --no-reference-mode 2>/dev/null | tail -1 > "$O/bench_under_rocprof.json"
bash tools/pmc_refresh.sh "$TAG" > "$O/pmc_refresh.log" 2>&1
timeout 400 python tools/stress_map.py 2>&1 | tail -1 > "$O/stress.json"
# BASELINE configs[2] / configs[3] on ONE GPU (the driver owns the 8-GPU runs), the host-scan hand-over, 4 pipelines per GPU
timeout 300 python bench.py --mode hypotheses --steps 60 2>/dev/null | tail -1 > "$O/bench_hypotheses.json"
SUMA_SEQ_CONCURRENT=2 timeout 400 python bench.py --mode sequences11 2>/dev/null | tail -1 > "$O/bench_sequences11.json"
timeout 300 python bench.py --mode adapter --adapter-scans 300 2>/dev/null | tail -1 > "$O/adapter_path_300_scans.json"
timeout 300 python tools/ingest_bench.py 2>/dev/null | tail -1 > "$O/ingest.json"
timeout 300 python tools/multi_seq.py 4 60 2>/dev/null | tail -1 > "$O/multi_seq.txt"
# N = 2 ranks started by bench.py itself (gloo, both on this box's one GPU), the per-block phase timeline, the
# Gauss-Newton station timeline
SUMA_BENCH_FORCE_DEVICE=0 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --cpu-scans 0 --adapter-scans 0 --no-kernel-events 2>/dev/null | tail -1 > "$O/bench_gpus2_self_launched_gloo.json"
timeout 300 python tools/phase_timeline.py 250 30 2>&1 | tail -19 > "$O/phase_timeline.txt"
timeout 300 python tools/gn_timeline.py 2>&1 | tail -14 > "$O/gn_timeline.txt"
# round 6: the loop-closure verification batched / serial / one guess, and the host-vector entry against the CPUs granted
timeout 300 python tools/loop_closure_timing.py 2048 150 33 > "$O/loop_closure_timing.json" 2>/dev/null
timeout 600 bash tools/host_entry_cpus.sh "$O/host_entry_cpus.jsonl" > /dev/null 2>&1
# LAST, so that it can never trail the sources again (round-4 review): BASELINE configs[1] verbatim, all 4541 scans of the
# shipped build against the oracle's recorded trace -- the JSON names the kernel sources (kernel_source_sha) it ran on
timeout 900 python tools/long_parity.py --check tests/golden/long_trace_4541.npz --out "$O/long_parity_4541_scans.json" 2>"$O/long_parity.err" | tail -1 | cut -c1-300
# summaries are made HERE and the raw counter / trace files dropped: gpurun copies back at most 64 MiB (round 6: the raw
# files of a full round no longer fit)
f1() { find "$O/$1" -name "$2" | head -1; }
python tools/rocprof_summary.py "$(f1 prof '*kernel_trace.csv')" > "$O/kernel_trace_summary.txt" 2>&1
cp "$(f1 prof '*kernel_stats.csv')" "$O/rocprofv3_kernel_stats.csv"
python tools/trace_gaps.py "$(f1 prof '*kernel_trace.csv')" 340 360 > "$O/scan_timeline_gaps.txt" 2>&1
python tools/make_hbm_traffic.py "$(f1 pmc_fetch '*counter_collection.csv')" "$(f1 pmc_write '*counter_collection.csv')" 2048 64 "$O/hbm_traffic.json" > "$O/hbm_traffic_pmc.txt" 2>&1
python tools/sq_summary.py "$(f1 pmc_sq1 '*counter_collection.csv')" "$(f1 pmc_sq2 '*counter_collection.csv')" "$(f1 pmc_sq3 '*counter_collection.csv')" > "$O/sq_summary.txt" 2>&1
rm -rf "$O/prof" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3"
du -sh "$O"
cat "$O/pytest_gpu.txt"; cut -c1-400 "$O/bench.json"; cut -c1-300 "$O/bench_full_sequence_4541.json"; ls "$O"
