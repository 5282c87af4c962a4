#!/bin/bash
# tools/ab_state.sh A.so B.so ... -- interleaved headline timing of builds on ONE box, every line with the box state bench.py reads off the
# telemetry (fast / slow, HISTORY.md section 7): a variant is judged in BOTH states
cd "$GRAFT_REPO_ROOT"
for rep in ${REPS:-1 2 3}; do
  for v in "$@"; do
    EMGPU_LIB=$PWD/$v python bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu-baseline --no-other-configs --no-host-path --verbose-line ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys, json
l = json.loads(sys.stdin.read()); r = l['roofline']; t = r.get('gpu_telemetry', {})
print('rep $rep %-24s %.3f ms  %s  %s W  %s MHz  %s' % ('$v', r['avg_step_ms'], l['config'].get('box_state'), t.get('socket_power_w', {}).get('median'), t.get('sclk_mhz', {}).get('median'), l['config']['kernel']))"
  done
done
