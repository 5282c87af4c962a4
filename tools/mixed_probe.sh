#!/bin/bash
# tools/mixed_probe.sh -- config 4 on ONE box: every v1.2 model alone at the rank's batch size, the mixed batch as one launch
# and as one launch per block.  Prints avg launch/step ms.
cd "$GRAFT_REPO_ROOT"
N=${N:-6250000}
one() { python bench.py --steps ${STEPS:-20} --warmup ${WARM:-10} --no-cpu-baseline --no-other-configs --no-host-path --verbose-line "$@" 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['roofline']['avg_step_ms'], l['roofline']['launches_per_step'], l['config']['kernel'])"; }
for m in uncor_1200exclude_fwme_v1p2 uncor_1200exclude_fwse_v1p2 uncor_1200exclude_rotorcraft_v1p2 uncor_1200only_fwme_v1p2 uncor_1200only_fwse_v1p2 uncor_1200only_rotorcraft_v1p2; do
  echo "$m alone n=$N: $(one --model $m --n $N)"
done
for rep in 1 2; do
  echo "mixed one launch: $(one --config mixed --n $N)"
  echo "mixed per block:  $(EMGPU_DEBUG_NO_MIXED_LAUNCH=1 one --config mixed --n $N)"
done
