#!/bin/bash
# Builds ablation variants of the convolution kernel (-DRP_ABL=n, see csrc/conv_igemm.hip) next to the real library.
#   bash tools/conv_ablate.sh 1 2 4 ...      (build container) -> gpurun_extra/abl_<n>.so (scratch: delete after the measurement, it ships with every gpurun lease)
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
mkdir -p $R/gpurun_extra
for n in "$@"; do
  /opt/rocm/bin/hipcc -c -x hip $R/rnnpose_amd/csrc/conv_igemm.hip -o $R/gpurun_extra/abl_$n.o -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize -DRP_ABL=$n &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/gpurun_extra/abl_$n.so $R/gpurun_extra/abl_$n.o $(ls $L/*.hip.o | grep -v conv_igemm) && echo built abl_$n
done
