for i in 1 2 3; do
  for L in tools/libsuma_hip_base.bin semantic_suma_amd/libsuma_hip.so; do
    SUMA_HIP_LIB=$L python bench.py --cpu-scans 0 --no-kernel-events --steady-scans 300 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), round(d['steady_state']['value'],1))"
  done
done
