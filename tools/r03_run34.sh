#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short 2>&1 | tail -4
bash tools/ab_lib.sh 3 rnnpose_amd/lib/sp_double.so
for lib in rnnpose_amd/lib/sp_double.so ""; do RNNPOSE_LIB=${lib:+$R/$lib} python bench.py --batch 1 --height 240 --width 240 --inner 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1 ${lib:-in-tree}', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; done
CONV_LAYERS_B=4,8 CONV_LAYERS_FILTER=3x3 python tools/conv_layers.py 0 f32 2>&1 | tail -10
RNNPOSE_LIB=$R/rnnpose_amd/lib/sp_double.so CONV_LAYERS_B=4,8 CONV_LAYERS_FILTER=3x3 python tools/conv_layers.py 0 f32 2>&1 | tail -10
