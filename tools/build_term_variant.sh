#!/bin/bash
# tools/build_term_variant.sh <name> [extra hipcc flags...] -- rebuild ONLY emgpu_kernels_term.hip with extra flags and link it with the
# in-tree objects into tools/ab/<name>.so (run with EMGPU_LIB=tools/ab/<name>.so; tools/ab_terminal.sh times several on one box).
#
# The measurement-only paths of that kernel -- the wave-level event queue (-DEMGPU_TERM_EVQ=N, profiles/r05_terminal_event_queue.txt), the
# ablation builds of HISTORY.md section 11.1 (-DEMGPU_TERM_ABL_LOADS / _ABL_ONEMODEL / _ABL_NOEVENTS / _ABL_NOFLUSH, -DEMGPU_TERM_NOSTORE,
# -DEMGPU_TERM_PLAIN_STORES) and the path counters (-DEMGPU_TERM_COUNTERS, tools/term_counters.py) -- are not in the product source since
# round 6: tools/patches/term_lab.patch puts them back into a COPY of csrc/ whenever one of those flags is given.  The patch was cut against
# the kernel as of the commit that introduced it; if the kernel has moved on since and the patch no longer applies, the round-5 tree
# (git 69dbb65, where the paths were in-tree) is checked out into /tmp and built instead: EMGPU_TERM_LAB_TREE=r05 forces that.
set -e
cd "$(dirname "$0")/.."
root=$(pwd)
name=$1; shift
src=em_model_manned_bayes_amd/csrc
mkdir -p tools/ab /tmp/emgpu_tv
HIPCC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --offload-arch=gfx950"
termsrc=$src/emgpu_kernels_term.hip
if echo " $* " | grep -q "EMGPU_TERM_\(EVQ\|ABL_\|NOSTORE\|PLAIN_STORES\|COUNTERS\)" || [ -n "$EMGPU_TERM_LAB_TREE" ]; then
    work=/tmp/emgpu_tv/lab_$name
    rm -rf $work; mkdir -p $work/em_model_manned_bayes_amd $work/include
    cp -r $src $work/em_model_manned_bayes_amd/csrc; cp include/emgpu.h $work/include/
    if [ "$EMGPU_TERM_LAB_TREE" != "r05" ] && (cd $work/$src && patch -p1 --quiet < $root/tools/patches/term_lab.patch); then
        termsrc=$work/$src/emgpu_kernels_term.hip
    else
        echo "term_lab.patch does not apply to the current kernel (or r05 was asked for): building the round-5 tree (git 69dbb65) instead" >&2
        rm -rf /tmp/emgpu_tv/r05; git worktree prune; git worktree add --detach /tmp/emgpu_tv/r05 69dbb65 >/dev/null
        make -C /tmp/emgpu_tv/r05/$src -j8 EXTRA="$*" >/dev/null
        cp /tmp/emgpu_tv/r05/em_model_manned_bayes_amd/libemgpu.so tools/ab/$name.so
        git worktree remove --force /tmp/emgpu_tv/r05
        ls -la tools/ab/$name.so
        exit 0
    fi
fi
$HIPCC "$@" -c $termsrc -o /tmp/emgpu_tv/$name.o
objs=$(ls $src/*.o | grep -v emgpu_kernels_term.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/$name.so $objs /tmp/emgpu_tv/$name.o
ls -la tools/ab/$name.so
