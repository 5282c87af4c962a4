#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] -- build the CURRENT csrc/ with extra flags into
# tools/ab/<name>.so (objects in a scratch directory; the in-tree library is untouched).  Run a variant with
# EMGPU_LIB=tools/ab/<name>.so, or time several on one box with tools/ab_bench.sh.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=em_model_manned_bayes_amd/csrc
obj=/tmp/emgpu_variant_$name
rm -rf $obj; mkdir -p $obj tools/ab
pids=()
for f in $src/*.cpp $src/*.hip; do
  b=$(basename $f); b=${b%.*}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --offload-arch=gfx950 "$@" -DEMGPU_SRC_HASH=\"variant-$name\" -c $f -o $obj/$b.o 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/$name.so $obj/*.o
ls -la tools/ab/$name.so
