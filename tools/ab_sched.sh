#!/bin/bash
# tools/ab_sched.sh -- the library built under LLVM's other AMDGPU scheduling strategies against the shipped build, every config of the bench line, one box
# (profiles/r06_sched_strategies.txt).  Build first:  bash tools/build_variant.sh sch_base;  bash tools/build_variant.sh sch_maxilp -mllvm -amdgpu-sched-strategy=max-ilp;
# bash tools/build_variant.sh sch_memclause -mllvm -amdgpu-sched-strategy=max-memory-clause
cd "$GRAFT_REPO_ROOT"
for cfg in uncor cor cor_v2p1_like mixed terminal; do
  echo "== $cfg"
  BENCH_ARGS="--config $cfg" STEPS=10 WARM=5 bash tools/ab_bench.sh tools/ab/sch_base.so tools/ab/sch_maxilp.so tools/ab/sch_memclause.so 2>&1 | grep -v amdgpu.ids
done
