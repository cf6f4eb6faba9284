#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -x -k "stem" 2>&1 | tail -5
timeout 600 python tools/bench_parity_probe.py 2>&1 | grep -v amdgpu.ids | tail -60
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; }
for i in 1 2; do ab RNNPOSE_LM_FUSED=0; ab RNNPOSE_LM_FUSED=1; done
for f in 0 1; do RNNPOSE_LM_FUSED=$f python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 --height 240 --width 240 --inner 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1 LM_FUSED=$f', d['value'], d['ms_per_step'])"; done
