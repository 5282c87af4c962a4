#!/bin/bash
# tools/build_term_variant.sh <name> [extra hipcc flags...] -- rebuild ONLY emgpu_kernels_term.hip with extra flags and link it with the
# in-tree objects into tools/ab/<name>.so (run with EMGPU_LIB=tools/ab/<name>.so; tools/ab_terminal.sh times several on one box)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=em_model_manned_bayes_amd/csrc
mkdir -p tools/ab /tmp/emgpu_tv
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --offload-arch=gfx950 "$@" -c $src/emgpu_kernels_term.hip -o /tmp/emgpu_tv/$name.o
objs=$(ls $src/*.o | grep -v emgpu_kernels_term.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/$name.so $objs /tmp/emgpu_tv/$name.o
ls -la tools/ab/$name.so
