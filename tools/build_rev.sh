#!/bin/bash
# builds libsuma_hip from a git revision (default HEAD) for A/B runs against the working tree:
#   tools/build_rev.sh <out.bin> [rev] [extra hipcc flags ...]
OUT=$(readlink -f "$1"); REV=${2:-HEAD}; shift; shift
D=$(mktemp -d)
git archive "$REV" semantic_suma_amd/csrc include | tar -x -C "$D" || exit 1
C=$D/semantic_suma_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -w"
SRC="k_preprocess k_filters k_icp k_render k_update suma_api suma_ingest k_sync suma_runner"
for f in $SRC; do ( cd $C && hipcc $FLAGS "$@" -c $f.hip -o $f.o ) & done
wait
( cd $C && hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $(for f in $SRC; do echo $f.o; done) -lpthread ) && echo "built $OUT from $REV"
rm -rf "$D"
