#!/bin/bash
# round-3 GPU pass 1: split-tensor convolution tests, per-layer microbench (all tile shapes), same-box bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r03a_conv_tests.txt; tail -5 gpurun_out/r03a_conv_tests.txt
timeout 300 python tools/conv_layers.py > gpurun_out/r03a_conv_layers.txt 2>&1; cat gpurun_out/r03a_conv_layers.txt
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > gpurun_out/r03a_pytest_gpu.log 2>&1; tail -8 gpurun_out/r03a_pytest_gpu.log
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r03a_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'conv frac', d['roofline']['frac'], 'chip', d['chip_level']['frac_of_fp16_mfma_peak'])"; }
for i in 1 2; do
ab RNNPOSE_SPLIT_TENSORS=0
ab RNNPOSE_SPLIT_TENSORS=1
ab RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=zr=3,zr2=3,heads=3,inp=3
ab RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=zr=3,zr2=3,q=3,q2=3,heads=3,inp=3,conv=3
ab RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=zr=2,zr2=2,heads=2,inp=2
done 2>&1 | tee gpurun_out/r03a_ab.txt
tail -3 gpurun_out/r03a_bench.err
