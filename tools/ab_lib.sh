#!/bin/bash
# Same-box A/B of builds of the library (box-to-box clocks differ by several per cent, so only runs on ONE box compare):
#   gpurun --timeout 600 -- 'bash tools/ab_lib.sh <rounds> gpurun_extra/a.so gpurun_extra/b.so ...'
# alternates bench.py over the given builds (RNNPOSE_LIB) and the in-tree build.
N=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for i in $(seq $N); do
  for lib in "$@" ""; do
    RNNPOSE_LIB=${lib:+$R/$lib} python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:-in-tree}', d['value'], 'iters/s', d['ms_per_step'], 'ms')"
  done
done | tee gpurun_out/ab_lib.log
