#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refiner.py -x -q -p no:cacheprovider --tb=short -k "encoder or refine or loop_S1 or golden or k_split or split" 2>&1 | tail -6
s1() { env "$@" python bench.py --batch 1 --height 240 --width 240 --inner 4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1 $*', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; }
for i in 1 2; do s1 RNNPOSE_ENCODER_MERGE_SMALL=1; s1 RNNPOSE_ENCODER_MERGE_SMALL=0; done
