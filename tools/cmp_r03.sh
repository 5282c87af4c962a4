cd "$GRAFT_REPO_ROOT"
STEPS=20 bash tools/ab_bench.sh tools/ab/r03 tools/ab/base.so
cd tools/ab/r03 && bash ../../../tools/pmc_cmd.sh k_uncor_fast "SQ_INSTS_VALU" bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1
