#!/bin/bash
# tools/ab_terminal.sh LIB... -- terminal propagation timing of several builds on one box (bench.py --config terminal)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in "$@"; do
  line=$(EMGPU_LIB=$PWD/$v python bench.py --config terminal --n 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "rep $rep $v $(echo "$line" | grep -o 'avg_step_ms": [0-9.]*')"
done; done
