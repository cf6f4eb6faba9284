#!/bin/bash
# Builds ablation variants of the convolution kernel (-DRP_ABL=n, see csrc/conv_igemm.hip) next to the real library.
#   bash tools/conv_ablate.sh 1 2 4 ...      (build container) -> rnnpose_amd/lib/abl_<n>.so
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
for n in "$@"; do
  /opt/rocm/bin/hipcc -c -x hip $R/rnnpose_amd/csrc/conv_igemm.hip -o $L/abl_$n.o -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize -DRP_ABL=$n &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/abl_$n.so $L/abl_$n.o $(ls $L/*.hip.o | grep -v conv_igemm) && echo built abl_$n
done
