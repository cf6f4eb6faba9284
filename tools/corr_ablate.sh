#!/bin/bash
# Builds diagnostics variants of the correlation-volume kernel next to the real library:
#   bash tools/corr_ablate.sh "name:-DRPC_ABL=1" "name2:-DRPC_NT=1 -DRPC_PF=2" ...   -> gpurun_extra/cv_<name>.so (scratch: delete after the measurement, it ships with every gpurun lease)
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
mkdir -p $R/gpurun_extra
for spec in "$@"; do
  n=${spec%%:*}; f=${spec#*:}
  /opt/rocm/bin/hipcc -c -x hip $R/rnnpose_amd/csrc/corr_pyramid.hip -o $R/gpurun_extra/cv_$n.o -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG $f &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/gpurun_extra/cv_$n.so $R/gpurun_extra/cv_$n.o $(ls $L/*.hip.o | grep -v corr_pyramid) && echo built cv_$n
done
