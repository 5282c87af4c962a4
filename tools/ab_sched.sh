cd "$GRAFT_REPO_ROOT"
for cfg in uncor cor cor_v2p1_like mixed terminal; do
  echo "== $cfg"
  BENCH_ARGS="--config $cfg" STEPS=10 WARM=5 bash tools/ab_bench.sh tools/ab/sch_base.so tools/ab/sch_maxilp.so tools/ab/sch_memclause.so 2>&1 | grep -v amdgpu.ids
done
