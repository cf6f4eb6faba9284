#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 300 python tools/conv_bench.py > $O/conv_bench2.log 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --kernel-include-regex "conv_igemm" --output-format csv -d $O/pmc_conv -o c -- python $R/tools/conv_bench.py > $O/pmc_conv.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace --kernel-include-regex "conv_igemm" --output-format csv -d $O/pmc_conv2 -o c -- python $R/tools/conv_bench.py > $O/pmc_conv2.log 2>&1
cd $R
cat $O/conv_bench2.log; ls $O/pmc_conv $O/pmc_conv2; tail -3 $O/pmc_conv.log $O/pmc_conv2.log
