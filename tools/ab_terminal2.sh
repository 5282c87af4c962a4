#!/bin/bash
# tools/ab_terminal2.sh A.so B.so ... -- like ab_terminal.sh, but prints the step time together with the track-seconds per encounter the
# build produced (ablation builds change the tracks: compare ms per track-second)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${N:-2000000}
for rep in 1 2 3; do
  for v in "$@"; do
    EMGPU_LIB=$PWD/$v timeout 120 python bench.py --config terminal --n $N --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-path --verbose-line --telemetry-s 0 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); ms=l['roofline']['avg_step_ms']; ts=l['config']['track_seconds_per_encounter']
print('rep $rep %-28s %.3f ms  %.1f track-s/enc  %.3f ns per track-second' % ('$v', ms, ts, ms*1e6/($N*ts)))"
  done
done
