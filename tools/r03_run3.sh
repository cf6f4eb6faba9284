#!/bin/bash
# round-3 GPU pass 3: deep-pipeline tile (4), RTN split, range-guard test diagnostics, parity numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short 2>&1 | tail -30 > gpurun_out/r03c_conv_tests.txt; tail -30 gpurun_out/r03c_conv_tests.txt
CONV_LAYERS_B=4,8,1 timeout 300 python tools/conv_layers.py 0 f32,hl1,hl4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03c_conv_layers.txt
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r03c_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'conv frac', d['roofline']['frac'], 'chip', d['chip_level']['frac_of_fp16_mfma_peak'])"; }
T4="convc2=4,convf2=4,conv=4,zr=4,q=4,zr2=4,q2=4,heads=4,inp=4"
for i in 1 2; do
ab RNNPOSE_SPLIT_TENSORS=0
ab RNNPOSE_SPLIT_TENSORS=1
ab RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=$T4
ab RNNPOSE_SPLIT_TENSORS=1 RNNPOSE_CONV_TILE=zr=4,q=4,zr2=4,q2=4
done 2>&1 | tee gpurun_out/r03c_ab.txt
tail -3 gpurun_out/r03c_bench.err
RNNPOSE_CONV_TILE=$T4 python bench.py --steps 5 --warmup 2 --cpu-runs 1 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench2.err; tail -c 300 gpurun_out/r03c_bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity'])
PY
