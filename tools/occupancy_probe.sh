#!/bin/bash
# tools/occupancy_probe.sh -- what a k_dbn_step2 instance loses with one resident wave less: the same launch with unused dynamic LDS
# (EMGPU_DEBUG_EXTRA_LDS) that takes a workgroup off every CU.  Round 4, one box: cor_v1 3 -> 2 waves 11.5 -> 14.4 ms, cor_v2p1_like
# 14.9 -> 17.7, glider_v1 4 -> 3 waves 11.4 -> 12.7 (-> 2: 16.9), uncor_1200code_v1 4 -> 3: 8.6 -> 9.5.
run() { EMGPU_DEBUG_EXTRA_LDS=$2 python bench.py --config $1 $3 --no-cpu-baseline --no-other-configs --no-host-path --steps 8 --warmup 3 --telemetry-s 0 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$1 $3 extra LDS $2: %.3f ms  %s' % (l['ms_per_step'], l['config']['kernel']))"; }
run cor 0; run cor 30000; run cor_v2p1_like 0; run cor_v2p1_like 30000
run uncor 0 "--model glider_v1"; run uncor 12000 "--model glider_v1"; run uncor 38000 "--model glider_v1"
run uncor 0 "--model uncor_1200code_v1"; run uncor 12000 "--model uncor_1200code_v1"
