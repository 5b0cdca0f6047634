#!/bin/bash
# quick A/B of two builds inside one gpurun session: GN parity tests on B, per-kernel table, three interleaved headline pairs
#   tools/ab_quick.sh <libA> <libB>
A=$1; B=$2
SUMA_HIP_LIB=$B python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_ref_golden.py -x -q -m gpu 2>&1 | tail -1
bash tools/ab_kernels.sh $A $B 2>&1 | tail -11
bash tools/ab.sh $A $B -- --no-loop-closure
