#!/bin/bash
# Diagnostics builds of the strip convolution kernel (-DRS_ABL=n, see csrc/conv_strip.hip) next to the real library.
#   bash tools/strip_ablate.sh 1 2 4 ...   (build container) -> gpurun_extra/sabl_<n>.so (scratch: delete after the measurement)
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
mkdir -p $R/gpurun_extra
# an argument vN builds schedule variant N (-DRS_VAR=N, results right) as sabl_100N.so
for n in "$@"; do
  D="-DRS_ABL=$n"; case $n in v*) D="-DRS_VAR=${n#v} $STRIP_DEFS"; n=100${n#v}$STRIP_SUFFIX;; esac
  /opt/rocm/bin/hipcc -c -x hip $R/rnnpose_amd/csrc/conv_strip.hip -o $R/gpurun_extra/sabl_$n.o -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize $D &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/gpurun_extra/sabl_$n.so $R/gpurun_extra/sabl_$n.o $(ls $L/*.hip.o | grep -v "/conv_strip.hip.o") && echo built sabl_$n &
done
wait
rm -f $R/gpurun_extra/sabl_*.o
