#!/bin/bash
# One-file variant of the library next to the real one (same-box A/B with tools/ab_lib.sh):
#   bash tools/variant_lib.sh <name> <file.hip> "-DFLAG=.. -DFLAG2=.."   -> gpurun_extra/<name>.so  (the other objects come from rnnpose_amd/lib)
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
NAME=$1; SRC=$2; DEFS=$3
mkdir -p $R/gpurun_extra
/opt/rocm/bin/hipcc -c -x hip $R/rnnpose_amd/csrc/$SRC -o $R/gpurun_extra/$NAME.o -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize $DEFS &&
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/gpurun_extra/$NAME.so $R/gpurun_extra/$NAME.o $(ls $L/*.hip.o | grep -v "/$SRC.o") && echo built $NAME
rm -f $R/gpurun_extra/$NAME.o
