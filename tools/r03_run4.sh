#!/bin/bash
# round-3 GPU pass 4: block-boundary overlap (barrier in the middle of the last tap) vs r02 boundary, tests, parity bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short 2>&1 | tail -30 > gpurun_out/r03d_conv_tests.txt; tail -12 gpurun_out/r03d_conv_tests.txt
for lib in novl librnnpose_hip; do echo "== $lib"; CONV_LAYERS_B=4,8,1 RNNPOSE_LIB=$R/rnnpose_amd/lib/$lib.so timeout 300 python tools/conv_layers.py 0 f32,hl1,hl4 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03d_conv_layers.txt
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/r03d_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'conv frac', d['roofline']['frac'], 'chip', d['chip_level']['frac_of_fp16_mfma_peak'])"; }
for i in 1 2; do
ab RNNPOSE_SPLIT_TENSORS=0 RNNPOSE_LIB=$R/rnnpose_amd/lib/novl.so
ab RNNPOSE_SPLIT_TENSORS=0
ab RNNPOSE_SPLIT_TENSORS=1
done 2>&1 | tee gpurun_out/r03d_ab.txt
tail -3 gpurun_out/r03d_bench.err
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x ) > gpurun_out/r03d_pytest_gpu.log 2>&1; tail -15 gpurun_out/r03d_pytest_gpu.log
RNNPOSE_SPLIT_TENSORS=0 python bench.py --steps 5 --warmup 2 --cpu-runs 1 > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench2.err; tail -c 300 gpurun_out/r03d_bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity'])
PY
