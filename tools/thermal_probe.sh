#!/bin/bash
# tools/thermal_probe.sh -- the headline launched by several processes on one box, with idle gaps: per-launch times, shader clock, socket
# power, temperatures and memory clock of each -- is a slow run the box warming up, and does --prewarm-s take it out of the line?
run() { python bench.py --no-cpu-baseline --no-other-configs --no-host-path --verbose-line --steps 20 --warmup 5 --prewarm-s $1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; t=r['gpu_telemetry']; print('prewarm $1: %.3f ms  sclk %s  W %s  temp %s  prewarm %s  steps %s' % (l['ms_per_step'], r['sclk_mhz'], t.get('socket_power_w',{}).get('median'), t.get('temperature_c'), r.get('prewarm'), [round(x,2) for x in r['step_ms'][:20]]))"; }
sleep 60; run 0; sleep 60; run 3; sleep 60; run 0; sleep 60; run 3
