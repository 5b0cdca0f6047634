#!/bin/bash
# A/B of one library under an environment switch inside ONE gpurun session:
#   tools/ab_env.sh VAR=value [bench args]   -> interleaved runs without / with the variable, prints the steady-state (value) and cold-start scans/s
SW="$1"; shift
for i in 1 2 3; do
  for E in "" "$SW"; do
    env $E python bench.py --cpu-scans 0 --no-kernel-events --adapter-scans 0 --no-host-vectors "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$E]', 'steady', round(d['value'],1), 'cold', round(d.get('cold_start',{}).get('value',0),1))"
  done
done
