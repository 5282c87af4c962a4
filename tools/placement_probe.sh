#!/bin/bash
# tools/placement_probe.sh -- does the headline's time depend on WHERE its buffers lie?  Processes one after the other on one box, the step's
# buffers pushed to other addresses by a dummy allocation ahead of them (EMGPU_BENCH_SHIFT_MB).
run() { EMGPU_BENCH_SHIFT_MB=$1 python bench.py --no-cpu-baseline --no-other-configs --no-host-path --verbose-line --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; t=r['gpu_telemetry']; print('shift %6s MB: %.3f ms  sclk %s  W %s' % ('$1', l['ms_per_step'], r['sclk_mhz'], t.get('socket_power_w',{}).get('median')))"; }
for rep in 1 2; do for s in 0 1000 3001 0 20000 70003 0 150000; do run $s; done; sleep 45; done
