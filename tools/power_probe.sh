#!/bin/bash
# tools/power_probe.sh [bench args] -- what the box does to its clocks while the benchmark kernel runs back to back:
# a long timed region (STEPS launches, default 1500) in the background, rocm-smi's power / clocks / temperature read a few
# times a second beside it (reading only: an ordinary user cannot change any of it).  Then the same launches with a pause
# between them (--step-gap-ms), which is how the profiler sees the kernel.  Output: gpurun_out/power_probe.log
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/power_probe.log
: > $out
sample() {   # $1 = pid to watch
  while kill -0 $1 2>/dev/null; do
    rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | python3 -c '
import json, sys, time
try:
    d = json.load(sys.stdin)
except Exception:
    sys.exit(0)
for card, v in d.items():
    keep = {k: x for k, x in v.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "fclk", "junction", "edge", "memory)"))}
    print(round(time.time(), 2), card, json.dumps(keep))
' >> $out
    sleep 0.2
  done
}
echo "== idle" >> $out
sleep 1 & sample $!
for gap in 0 ${GAP_MS:-3}; do
  echo "== back to back launches, gap ${gap} ms" >> $out
  timeout 300 python3 bench.py --steps ${STEPS:-1500} --warmup 5 --no-cpu-baseline --no-other-configs --no-host-path --step-gap-ms $gap "$@" > gpurun_out/power_probe_bench_$gap.json 2>gpurun_out/power_probe_bench_$gap.err &
  pid=$!
  sample $pid
  wait $pid
  tail -1 gpurun_out/power_probe_bench_$gap.json | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("line:", d["ms_per_step"], "ms/step; kernel", d["roofline"]["avg_launch_ms"], "ms (HIP events)", d["roofline"]["frac"])' >> $out
done
tail -50 $out
