#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python tools/error_budget.py 2>&1 | grep -v amdgpu.ids | tail -8
export CONV_LAYERS_FILTER="zr 1x5,heads,convc2" CONV_LAYERS_B=4,8
for n in 0 32 1; do
  lib=$R/rnnpose_amd/lib/abl_$n.so; [ $n = 0 ] && lib=$R/rnnpose_amd/lib/librnnpose_hip.so
  echo "== RP_ABL=$n (32: half of the waves request weights; 1: none)"
  RNNPOSE_LIB=$lib timeout 120 python tools/conv_layers.py 0 f32,hl1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03f_ablate_halfweights.txt
for shape in "--batch 16 --height 240 --width 240 --inner 4" "--batch 32 --height 240 --width 240 --inner 4" "--batch 1 --height 240 --width 240 --inner 4"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $shape 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$shape', d['value'], 'iters/s', d['ms_per_step'], 'ms  chip', d['chip_level']['frac_of_fp16_mfma_peak'], 'conv', d['roofline']['frac'] if d['roofline'] else None)"
done | tee gpurun_out/r03f_shapes.txt
