#!/bin/bash
# copies the evidence of gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into profiles/ with
# the round prefix and rebuilds the summaries bench.py / the judge read:  bash tools/collect_profiles.sh r03 [suffix]
TAG=${1:?tag}; SFX=${2:-}; O=gpurun_out/$TAG; P=profiles/${TAG}${SFX}
cp "$O/pytest_gpu.txt" "${P}_pytest_gpu.txt"
cp "$O/bench.json" "${P}_bench.json"
cp "$O/bench_kernels_hip_events.json" "${P}_bench_kernels_hip_events.json"
cp "$O/bench_under_rocprof.json" "${P}_bench_under_rocprof.json"
# round 6: the summaries are made on the GPU box (tools/profile_round.sh); older rounds made them here from the raw files
if [ -s "$O/rocprofv3_kernel_stats.csv" ]; then
  cp "$O/rocprofv3_kernel_stats.csv" "${P}_rocprofv3_kernel_stats.csv"
  cp "$O/kernel_trace_summary.txt" "${P}_kernel_trace_summary.txt"
  cp "$O/hbm_traffic_pmc.txt" "${P}_hbm_traffic_pmc.txt"; cp "$O/hbm_traffic.json" profiles/hbm_traffic.json
  cp "$O/sq_summary.txt" "${P}_sq_summary.txt"
else
  cp "$O"/prof/*kernel_stats.csv "${P}_rocprofv3_kernel_stats.csv"
  python tools/rocprof_summary.py "$O"/prof/*kernel_trace.csv > "${P}_kernel_trace_summary.txt"
  python tools/make_hbm_traffic.py "$O"/pmc_fetch/*counter_collection.csv "$O"/pmc_write/*counter_collection.csv 2048 64 profiles/hbm_traffic.json > "${P}_hbm_traffic_pmc.txt"
  python tools/sq_summary.py "$O"/pmc_sq1/*counter_collection.csv "$O"/pmc_sq2/*counter_collection.csv "$O"/pmc_sq3/*counter_collection.csv > "${P}_sq_summary.txt"
fi
cp "$O/stress.json" "${P}_stress_50M_surfels_128x4096.json"
for f in long_parity_4541_scans.json bench_driver_shape_steps20.json bench_full_sequence_4541.json bench_hypotheses.json bench_sequences11.json adapter_path_300_scans.json ingest.json multi_seq.txt bench_gpus2_self_launched_gloo.json phase_timeline.txt gn_timeline.txt loop_closure_timing.json host_entry_cpus.jsonl scan_timeline_gaps.txt; do [ -s "$O/$f" ] && cp "$O/$f" "${P}_$f"; done
ls -la profiles | tail -24
