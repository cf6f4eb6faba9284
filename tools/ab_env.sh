#!/bin/bash
# Same-box A/B of runtime switches:  bash tools/ab_env.sh <rounds> "VAR=1 VAR2=0" "VAR=0" ...   ("" = defaults)
N=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in $(seq $N); do
  for e in "$@"; do
    env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[${e:-defaults}]', d['value'], 'iters/s', d['ms_per_step'], 'ms')"
  done
done | tee gpurun_out/ab_env.log
