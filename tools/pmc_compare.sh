#!/bin/bash
# tools/pmc_compare.sh -- SQ counters of the headline kernel on two models side by side (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_INSTS_VALU[A-Z_0-9]*\|SQ_WAIT_INST[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_THREAD_CYCLES[A-Z_]*\|SQ_VALU[A-Z_]*" | sort -u | tr '\n' ' '
echo
for m in "" "--model uncor_1200only_fwse_v1p2"; do
  bash tools/pmc_quick.sh "$m" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
  bash tools/pmc_quick.sh "$m" SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR
  bash tools/pmc_quick.sh "$m" SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_WAVES
done
