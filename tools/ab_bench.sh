#!/bin/bash
# tools/ab_bench.sh libA.so libB.so [...] -- interleaved timing of library builds on ONE box
# (devices differ by several percent: never compare numbers from different gpurun calls).
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for lib in "$@"; do
    ms=$(EMGPU_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | grep -o 'avg_launch_ms": [0-9.]*' | cut -d' ' -f2)
    echo "rep $rep $lib $ms"
  done
done
