#!/bin/bash
# builds a variant of libsuma_hip.so for A/B runs: tools/build_variant.sh <out.so> [extra hipcc flags ...]
# (objects go to a scratch directory; the product build in semantic_suma_amd/csrc is not touched)
OUT=$1; shift
D=$(mktemp -d); C=semantic_suma_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -w"
SRC="k_preprocess k_filters k_icp k_render k_update suma_api suma_ingest k_sync suma_runner"
pids=()
for f in $SRC; do
  hipcc $FLAGS "$@" -c $C/$f.hip -o $D/$f.o &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
for f in $SRC; do [ -s $D/$f.o ] || { echo "build_variant: $f.hip did not compile" >&2; fail=1; }; done
[ $fail -eq 0 ] || { rm -rf $D; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $D/*.o -lpthread && rm -rf $D && echo "built $OUT"
