#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > gpurun_out/r03n_pytest_gpu.log 2>&1; tail -12 gpurun_out/r03n_pytest_gpu.log
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r03n_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'conv frac', d['roofline']['frac'], 'chip', d['chip_level']['frac_of_fp16_mfma_peak'])"; }
for i in 1 2; do ab RNNPOSE_SPATIAL_TILES=0; ab RNNPOSE_SPATIAL_TILES=1; ab RNNPOSE_SPATIAL_TILES=1 RNNPOSE_SPLIT_TENSORS=1; done 2>&1 | tee gpurun_out/r03n_ab.txt
for sp in 0 1; do RNNPOSE_SPATIAL_TILES=$sp python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 --height 240 --width 240 --inner 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1 spatial=$sp', d['value'], d['ms_per_step'])"; done
