#!/bin/bash
# tools/profile_bench.sh TAG -- run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of
# the default bench.py command, then the HBM counters in their own passes (WRITE_SIZE and FETCH_SIZE
# cannot share a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then SQ counters.
# Every pass is bounded by `timeout` (an unsupported counter can hang the profiler).  Raw output lands in gpurun_out/prof_$TAG; tools/summarize_profiles.py condenses it into profiles/.
set -u
TAG=${1:-r04}   # bench.py defaults: --warmup 5 --steps 10
shift || true
EXTRA="$*"      # further bench.py arguments, e.g. --config cor (the tag should then say so: r02_cor)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $EXTRA --no-cpu-baseline --no-other-configs --no-host-path --telemetry-s 0 > $OUT/bench_kt.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $EXTRA --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --telemetry-s 0 --prewarm-s 0 --placement-candidates 1 > $OUT/bench_pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $EXTRA --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --telemetry-s 0 --prewarm-s 0 --placement-candidates 1 > $OUT/bench_pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py $EXTRA --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --telemetry-s 0 --prewarm-s 0 --placement-candidates 1 > $OUT/bench_pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $EXTRA --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --telemetry-s 0 --prewarm-s 0 --placement-candidates 1 > $OUT/bench_pmc_sq2.log 2>&1
timeout 600 python3 bench.py $EXTRA ${PLAIN_EXTRA:-} > $OUT/bench_plain.json 2> $OUT/bench_plain.err
tail -1 $OUT/bench_plain.json | cut -c1-400
