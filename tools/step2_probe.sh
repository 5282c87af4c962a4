#!/bin/bash
# tools/step2_probe.sh [dir ...] -- the per-timestep kernel's workloads on ONE box: the working tree, then every built checkout given
# (tools/ab_checkout.sh <commit> <name> makes tools/ab/<name>/)
cd "$GRAFT_REPO_ROOT"
ROOT=$PWD
one() { timeout 120 python bench.py --steps ${STEPS:-10} --warmup ${WARM:-5} --no-cpu-baseline --no-other-configs --no-host-path --verbose-line "$@" 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('%.3f ms  frac %.3f  %s' % (l['roofline']['avg_step_ms'], l['roofline']['frac'], l['config']['kernel']))"; }
for rep in 1 2; do
for d in . "$@"; do
  cd $ROOT/$d
  echo "[$d] cor_v1:        $(one --config cor)"
  echo "[$d] cor_v2p1_like: $(one --config cor --model cor_v2p1_like)"
  echo "[$d] glider_v1:     $(one --model glider_v1)"
  echo "[$d] per-step v2p1: $(one --per-step)"
  echo "[$d] 1200code_v1:   $(one --model uncor_1200code_v1)"
done
done
