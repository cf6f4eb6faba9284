#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short 2>&1 | tail -12
python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider --tb=short -k "encoder or update or stem or instnorm" 2>&1 | tail -8
for sp in 1 0; do echo "== RNNPOSE_SPATIAL_TILES=$sp"; RNNPOSE_SPATIAL_TILES=$sp CONV_LAYERS_FILTER="3x3" CONV_LAYERS_B=4,8,1 timeout 300 python tools/conv_layers.py 0 f32,hl0 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03l_spatial_layers.txt
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r03l_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'conv frac', d['roofline']['frac'], 'chip', d['chip_level']['frac_of_fp16_mfma_peak'])"; }
for i in 1 2; do ab RNNPOSE_SPATIAL_TILES=0; ab RNNPOSE_SPATIAL_TILES=1; done 2>&1 | tee gpurun_out/r03l_ab.txt
tail -3 gpurun_out/r03l_bench.err
