#!/bin/bash
# A/B builds of libsuma_hip.so inside ONE gpurun session (box-to-box variance is ~5 %):
#   tools/ab.sh <libA.so> <libB.so> [more .so ...] [-- bench args]   -> interleaved runs, prints scans/s
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in 1 2 3; do
  for L in "${LIBS[@]}"; do
    v=$(SUMA_HIP_LIB=$L python bench.py --cpu-scans 0 --no-kernel-events --adapter-scans 0 --no-host-vectors "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    echo "$L $v"
  done
done
