#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_refiner.py -x -q -p no:cacheprovider --tb=short -k "k_split" 2>&1 | tail -8
CONV_LAYERS_B=1 python tools/conv_layers.py 0 f32,f32ks 2>&1 | tail -12
s1() { env "$@" python bench.py --batch 1 --height 240 --width 240 --inner 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1 $*', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; }
for i in 1 2; do s1 RNNPOSE_KSPLIT=1; s1 RNNPOSE_KSPLIT=0; done
