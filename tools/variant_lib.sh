#!/bin/bash
# Variant of the library next to the real one (same-box A/B with tools/ab_lib.sh): the named files (comma-separated) are compiled with extra
# definitions, the other objects come from rnnpose_amd/lib:
#   bash tools/variant_lib.sh <name> <file.hip[,file2.hip,...]> "-DFLAG=.. -DFLAG2=.."   -> gpurun_extra/<name>.so
R=$(cd $(dirname $0)/.. && pwd)
L=$R/rnnpose_amd/lib
NAME=$1; SRCS=$2; DEFS=$3
mkdir -p $R/gpurun_extra
OBJS=""; SKIP=""
for SRC in ${SRCS//,/ }; do
  /opt/rocm/bin/hipcc -c -x hip $R/rnnpose_amd/csrc/$SRC -o $R/gpurun_extra/$NAME.$SRC.o -I $R/include -I $R/rnnpose_amd/csrc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNDEBUG -fno-slp-vectorize $DEFS || exit 1
  OBJS="$OBJS $R/gpurun_extra/$NAME.$SRC.o"; SKIP="$SKIP -e /$SRC.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/gpurun_extra/$NAME.so $OBJS $(ls $L/*.hip.o | grep -v $SKIP) && echo built $NAME
rm -f $OBJS
