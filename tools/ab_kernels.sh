#!/bin/bash
# per-kernel A/B inside ONE gpurun session: bench.py's untimed per-kernel HIP-event table for two builds
#   tools/ab_kernels.sh <libA> <libB>  -> gpurun_out/abk_A.json, abk_B.json + a side-by-side table
A=$1; B=$2
for i in 1 2; do
  SUMA_HIP_LIB=$A python bench.py --cpu-scans 0 --adapter-scans 0 --no-host-vectors --profile-scans 40 --kernels-json gpurun_out/abk_A$i.json 2>/dev/null | tail -1 | cut -c1-80
  SUMA_HIP_LIB=$B python bench.py --cpu-scans 0 --adapter-scans 0 --no-host-vectors --profile-scans 40 --kernels-json gpurun_out/abk_B$i.json 2>/dev/null | tail -1 | cut -c1-80
done
python - <<'PY'
import json
def load(p):
    return {k["name"]: k for k in json.load(open(p))["kernels"]}
A = [load(f"gpurun_out/abk_A{i}.json") for i in (1, 2)]
B = [load(f"gpurun_out/abk_B{i}.json") for i in (1, 2)]
avg = lambda X, n: sum(x[n]["avg_us"] for x in X if n in x) / max(1, sum(n in x for x in X))
for n in A[0]:
    print(f"{n:28s} A {avg(A, n):8.2f} us   B {avg(B, n):8.2f} us   {100.0 * (avg(B, n) / avg(A, n) - 1.0):+6.1f} %")
PY
