#!/bin/bash
# Diagnostics builds for the kernel-to-kernel visibility experiments (r05, profiles/r05_determinism.txt), next to the real library:
#   gpurun_extra/det_<name>.so, selected with RNNPOSE_LIB.   bash tools/det_variants.sh   (build container; run the normal build first)
#   mu_old      mask_upsample with the r02-r04 epilogue (4-byte stores straight from registers)
#   mu_old_acq  + corr_weight starts with an agent-scope acquire (buffer_inv sc1)
#   mu_old_sc1  + corr_weight reads the flow map with sc1 loads
#   acq         the in-tree mask_upsample (whole-line stores) + the acquire in corr_weight
#   wt          in-tree mask_upsample with WRITE-THROUGH stores (global_store_dwordx4 sc0 sc1) + every wave waits for them
#   rel         in-tree mask_upsample + an agent-scope release per workgroup
#   pw_slp      pointwise.hip (corr_weight) compiled as until r04: plain -O3, SLP vectoriser on -> packed fp32 instructions.  THE variant
#               that reproduces r04's finding; every other build here has -fno-slp-vectorize like the library itself
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
X=$R/gpurun_extra
mkdir -p $X
CC="/opt/rocm/bin/hipcc -c -x hip -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize"
LD="/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950"
others() { ls $L/*.hip.o | grep -v -e "/mask_upsample.hip.o" -e "/pointwise.hip.o"; }
$CC $R/rnnpose_amd/csrc/mask_upsample.hip -DMU_LINE_STORES=0 -o $X/mu_old.o &
$CC $R/rnnpose_amd/csrc/mask_upsample.hip -DMU_STORE_WT=1 -o $X/mu_wt.o &
$CC $R/rnnpose_amd/csrc/mask_upsample.hip -DMU_RELEASE=1 -o $X/mu_rel.o &
$CC $R/rnnpose_amd/csrc/pointwise.hip -DRP_CW_ACQUIRE=1 -o $X/pw_acq.o &
$CC $R/rnnpose_amd/csrc/pointwise.hip -fslp-vectorize -o $X/pw_slp.o &
$CC $R/rnnpose_amd/csrc/pointwise.hip -DRP_CW_SC1=1 -o $X/pw_sc1.o &
wait
$LD -o $X/det_mu_old.so $X/mu_old.o $L/pointwise.hip.o $(others) &&
$LD -o $X/det_mu_old_acq.so $X/mu_old.o $X/pw_acq.o $(others) &&
$LD -o $X/det_mu_old_sc1.so $X/mu_old.o $X/pw_sc1.o $(others) &&
$LD -o $X/det_wt.so $X/mu_wt.o $L/pointwise.hip.o $(others) &&
$LD -o $X/det_rel.so $X/mu_rel.o $L/pointwise.hip.o $(others) &&
$LD -o $X/det_wt_acq.so $X/mu_wt.o $X/pw_acq.o $(others) &&
$LD -o $X/det_pw_slp.so $L/mask_upsample.hip.o $X/pw_slp.o $(others) &&
$LD -o $X/det_acq.so $L/mask_upsample.hip.o $X/pw_acq.o $(others) && echo built 8 variants
rm -f $X/*.o
