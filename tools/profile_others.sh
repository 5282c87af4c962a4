#!/bin/bash
# tools/profile_others.sh TAG -- rocprofv3 kernel trace + stats of the non-headline workloads (one GPU), every pass bounded by `timeout`:
# the v1.2 family on k_uncor_fast<7,4,6,6>, the mixed batch in one launch, the per-timestep kernel on cor_v1 / cor_v2p1_like / glider_v1 /
# uncor_1200code_v1 / PER_STEP / littoral_cor_v1 (frozen columns), the event-list kernels, the sample2track consumer, the
# UncorEncounterModel.track pipeline and terminal propagation.  tools/summarize_others.py condenses the stats.
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/others_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { # name, program args...
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1
}
B="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-host-path"
run v1p2 bench.py $B --model uncor_1200only_fwse_v1p2
run mixed bench.py $B --config mixed
run cor_v1 bench.py $B --config cor
run cor_v2p1_like bench.py $B --config cor_v2p1_like
run glider_v1 bench.py $B --model glider_v1
run uncor_1200code_v1 bench.py $B --model uncor_1200code_v1
run per_step bench.py $B --per-step
run littoral_cor_v1 bench.py $B --model littoral_cor_v1
run events tools/bench_events.py uncor_1200code_v2p1 uncor_1200only_fwse_v1p2 uncor_1200code_v1 glider_v1 cor_v1 littoral_cor_v1
run track tools/bench_track.py 4000000 240
run utrack tools/bench_utrack.py 1000000 240
run terminal bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --config terminal --n 1000000
tail -n 6 $OUT/events.log; tail -n 1 $OUT/track.log $OUT/utrack.log
