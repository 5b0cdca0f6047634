#!/bin/bash
# builds a variant of libsuma_hip.so for A/B runs: tools/build_variant.sh <out.so> [extra hipcc flags ...]
# (objects go to a scratch directory; the product build in semantic_suma_amd/csrc is not touched)
OUT=$1; shift
D=$(mktemp -d); C=semantic_suma_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -w"
for f in k_preprocess k_filters k_icp k_render k_update suma_api suma_ingest k_sync suma_runner; do
  hipcc $FLAGS "$@" -c $C/$f.hip -o $D/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $D/*.o -lpthread && rm -rf $D && echo "built $OUT"
