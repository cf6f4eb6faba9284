#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for lib in librnnpose_hip c4occ2; do echo "== $lib"; CONV_LAYERS_FILTER="zr 1x5,q 1x5,heads,convc2,conv 3x3" CONV_LAYERS_B=8,4 RNNPOSE_LIB=$R/rnnpose_amd/lib/$lib.so timeout 300 python tools/conv_layers.py 0 f32,hl1,hl2 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03k_cols4.txt
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; }
T2="convc2=2,conv=2,zr=2,q=2,zr2=2,q2=2,heads=2,inp=2"
ab RNNPOSE_X=0
ab RNNPOSE_SPLIT_BATCH=0
ab RNNPOSE_SPLIT_BATCH=0 RNNPOSE_SPLIT_TENSORS=1
ab RNNPOSE_SPLIT_BATCH=0 RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=$T2 RNNPOSE_LIB=$R/rnnpose_amd/lib/c4occ2.so
ab RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=$T2 RNNPOSE_LIB=$R/rnnpose_amd/lib/c4occ2.so
