#!/bin/bash
# tools/profile_others.sh TAG -- rocprofv3 kernel trace + stats of the non-headline workloads (one GPU):
# the v1.2 family on k_uncor_fast<7,4,6,6>, the per-timestep kernel on cor_v1 / glider_v1 / PER_STEP,
# the sample2track consumer and terminal propagation.  tools/summarize_others.py condenses the stats.
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/others_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1
}
run v1p2 bench.py --steps 5 --warmup 2 --no-cpu-baseline --model uncor_1200only_fwse_v1p2
run mixed bench.py --steps 5 --warmup 2 --no-cpu-baseline --config mixed
run cor_v1 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cor
run cor_v2p1_like bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cor --model cor_v2p1_like
run glider_v1 bench.py --steps 5 --warmup 2 --no-cpu-baseline --model glider_v1
run per_step bench.py --steps 5 --warmup 2 --no-cpu-baseline --per-step
run track tools/bench_track.py 4000000 240
run utrack tools/bench_utrack.py 1000000 240
run terminal bench.py --steps 3 --warmup 1 --no-cpu-baseline --config terminal --n 1000000
tail -n 1 $OUT/track.log $OUT/utrack.log
