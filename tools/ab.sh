#!/bin/bash
# A/B two builds of libsuma_hip.so inside ONE gpurun session (box-to-box variance is ~5 %):
#   tools/ab.sh <libA.so> <libB.so> [bench args]   -> interleaved runs, prints scans/s
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    v=$(SUMA_HIP_LIB=$L python bench.py --cpu-scans 0 --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    echo "$L $v"
  done
done
