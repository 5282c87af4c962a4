#!/bin/bash
# tools/ab_terminal.sh A.so B.so ... -- interleaved timing of k_terminal_propagate builds on ONE box (1 M encounters)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for v in "$@"; do
    ms=$(EMGPU_LIB=$PWD/$v timeout 120 python bench.py --config terminal --n 1000000 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['roofline']['avg_step_ms'])")
    echo "rep $rep $v $ms"
  done
done
