#!/bin/bash
# Same-box A/B of runtime switches at the reference's own working size (B = 1, 240 x 240, 3 x 4):  bash tools/ab_s1.sh <rounds> "VAR=1" "" ...
N=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in $(seq $N); do
  for e in "$@"; do
    env $e python bench.py --batch 1 --height 240 --width 240 --outer 3 --inner 4 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[S1 ${e:-defaults}]', d['ms_per_step'], 'ms')"
  done
done | tee gpurun_out/ab_s1.log
