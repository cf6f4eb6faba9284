#!/bin/bash
# times the strip-kernel diagnostics builds (tools/strip_ablate.sh) on a few update-block layers:  gpurun -- 'bash tools/strip_ablate_run.sh <tag>'
TAG=${1:-s}
for lib in rnnpose_amd/lib/librnnpose_hip $(ls gpurun_extra | grep -E "^sabl_[0-9a-z_]+\.so$" | sed 's/\.so//' | sort -t_ -k2 -n | sed 's#^#gpurun_extra/#'); do
  echo "== $lib"
  CONV_LAYERS_FILTER="${STRIP_LAYERS:-zr 1x5,q 1x5,heads,convc2}" CONV_LAYERS_B=${STRIP_B:-4,8} RNNPOSE_LIB=$PWD/$lib.so timeout 120 python tools/conv_layers.py 0 ${STRIP_MODES:-hl5,f32t5} 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${TAG}_strip_ablation.txt 2>&1
cat gpurun_out/${TAG}_strip_ablation.txt
