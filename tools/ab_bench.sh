#!/bin/bash
# tools/ab_bench.sh A B [...] -- interleaved timing of builds on ONE box (devices differ by several
# percent: never compare numbers from different gpurun calls).  An argument is either a built
# libemgpu.so (run with the current Python side through EMGPU_LIB) or a directory holding a built
# checkout of another commit (tools/ab_checkout.sh <commit> <name> makes tools/ab/<name>/).
cd "$GRAFT_REPO_ROOT"
ROOT=$PWD
for rep in 1 2 3; do
  for v in "$@"; do
    if [ -d "$ROOT/$v" ]; then
      line=$(cd "$ROOT/$v" && python bench.py --steps ${STEPS:-10} --warmup ${WARM:-5} --no-cpu-baseline --no-other-configs --no-host-path ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    else
      line=$(EMGPU_LIB=$ROOT/$v python bench.py --steps ${STEPS:-10} --warmup ${WARM:-5} --no-cpu-baseline --no-other-configs --no-host-path ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    fi
    ms=$(echo "$line" | grep -o 'avg_launch_ms": [0-9.]*' | head -1 | cut -d' ' -f2)
    kn=$(echo "$line" | grep -o '"kernel": "[^"]*"' | head -1 | cut -d'"' -f4)
    echo "rep $rep $v $ms $kn"
  done
done
