#!/bin/bash
# tools/ld_probe.sh -- how the trace's leading dimension (row alignment) moves the dense kernels, on ONE box
cd "$GRAFT_REPO_ROOT"
one() { python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-configs --no-host-path --verbose-line "$@" 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['roofline']['avg_step_ms'], l['config']['kernel'])"; }
for rep in 1 2; do
for pad in 1 256 1024 4096 16384; do
  echo "uncor 10M pad $pad: $(one --ld-pad $pad)"
done
done
for pad in 1 1024 16384; do
  echo "mixed 6.25M pad $pad: $(one --config mixed --ld-pad $pad)"
  echo "cor 10M pad $pad: $(one --config cor --ld-pad $pad)"
done
