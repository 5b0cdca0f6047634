#!/bin/bash
# A/B of one library under an environment switch inside ONE gpurun session:
#   tools/ab_env.sh VAR=value [bench args]   -> interleaved runs without / with the variable, prints scans/s and steady state
SW="$1"; shift
for i in 1 2 3; do
  for E in "" "$SW"; do
    env $E python bench.py --cpu-scans 0 --no-kernel-events --steady-scans 300 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$E]', round(d['value'],1), round(d['steady_state']['value'],1))"
  done
done
