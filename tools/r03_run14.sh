#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short 2>&1 | tail -6
for sp in 1 0; do echo "== RNNPOSE_SPATIAL_TILES=$sp"; RNNPOSE_SPATIAL_TILES=$sp CONV_LAYERS_FILTER="heads,convc2,conv 3x3" CONV_LAYERS_B=4,8 timeout 300 python tools/conv_layers.py 0 f32t1,f32t2,hl1,hl2 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03m_heads.txt
