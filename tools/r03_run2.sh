#!/bin/bash
# round-3 GPU pass 2: ablation builds of the convolution kernel (which part of a tile's life is the time?), bench parity block
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export CONV_LAYERS_FILTER="zr 1x5,q 1x5,heads,enc l1" CONV_LAYERS_B=4,1
for n in 0 1 2 4 8 16 3 7 15 31; do
  lib=$R/rnnpose_amd/lib/abl_$n.so; [ $n = 0 ] && lib=$R/rnnpose_amd/lib/librnnpose_hip.so
  echo "== RP_ABL=$n (1 no weight loads, 2 no LDS fragment reads, 4 no activation staging, 8 no barrier, 16 no epilogue stores)"
  RNNPOSE_LIB=$lib timeout 120 python tools/conv_layers.py 0 f32,hl1,hl3 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03b_ablate.txt
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider -k "range_guard or split" 2>&1 | tail -4
python bench.py --steps 10 --warmup 2 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; tail -c 600 gpurun_out/r03b_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity'], d['cpu_baseline']['value'], d['cpu_baseline']['min'], d['cpu_baseline']['max'], d['f16x3_range_events'])
PY
