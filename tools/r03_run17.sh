#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short -k "range or saturation" 2>&1 | tail -4
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; }
for i in 1 2 3; do ab RNNPOSE_RANGE_GUARD=1; ab RNNPOSE_RANGE_GUARD=0; done
