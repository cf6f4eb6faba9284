#!/bin/bash
# SQ / TCC counter passes (rocprofv3 --pmc, one pass per counter set, --kernel-trace only -- never combined with
# --stats/--sys-trace) over a command; results stay as rocprofv3 databases under gpurun_out/<tag>_pmc<i>/.
#   gpurun --timeout 900 -- 'bash tools/pmc_sq.sh <tag> python tools/pmc_kernels.py 2'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F16" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$PMC_TRAFFIC_ONLY" ] && [ $i -lt 3 ]; then continue; fi     # FETCH_SIZE / WRITE_SIZE passes only (pmc3, pmc4)
  if [ -n "$PMC_SQ_ONLY" ] && [ $i -gt 2 ]; then continue; fi          # the two SQ passes only
  ( cd $R && rocprofv3 --pmc $SET --kernel-trace -d $OUT/${TAG}_pmc$i -o run -- "$@" ) > $OUT/${TAG}_pmc$i.log 2>&1
  tail -1 $OUT/${TAG}_pmc$i.log | cut -c1-200
done
ls $OUT | grep ${TAG}_pmc
