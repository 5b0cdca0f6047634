#!/bin/bash
# round-5 GPU session 2: parity of the new arithmetic specification (whole GPU suite), driver-shaped bench
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s2; mkdir -p "$O"
export SUMA_SCAN_CACHE=/tmp/suma_scans
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > "$O/pytest_gpu.txt"; tail -12 "$O/pytest_gpu.txt"
timeout 300 python bench.py --steps 20 2>"$O/bench_driver.err" | tail -1 > "$O/bench_driver.json"; tail -16 "$O/bench_driver.err"; cut -c1-200 "$O/bench_driver.json"
