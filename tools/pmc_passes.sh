#!/bin/bash
# Collect SQ/TA/TCP counters for the contraction kernel, one small counter group per rocprofv3 pass
# (counters only: --pmc is never combined with a trace domain).  usage: pmc_passes.sh OUTDIR [target args]
# PMC_TARGET=bench_split.py: the split-state contraction on the 16->64 and 64->256 shapes instead (round 3)
# (TA_*/TCP_* derived sums are left out: that pass did not finish within 180 s on this pool.)
# Stops at the first pass that times out or is killed.
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $group -d "$out/p$i" -o p --output-format csv -- \
      python3 "$GRAFT_REPO_ROOT/tools/${PMC_TARGET:-pmc_target.py}" "$@" > "$out/p$i.log" 2>&1
  rc=$?
  echo "pass $i [$group] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping: pass killed"; exit 1; fi
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_VMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_IFETCH SQ_INST_CYCLES_SALU
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY
GROUPS
