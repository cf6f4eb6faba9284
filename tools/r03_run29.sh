#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
python bench.py --batch 1 --height 240 --width 240 --inner 4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1', d['value'], 'iters/s', d['ms_per_step'], 'ms')"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/s1_prof -o run -- python $R/bench.py --batch 1 --height 240 --width 240 --inner 4 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/s1_prof.log 2>&1
cd $R
python tools/timeline.py $(ls $OUT/s1_prof/*/run_results.db $OUT/s1_prof/run_results.db 2>/dev/null | head -1) > $OUT/s1_timeline.txt 2>&1
cat $OUT/s1_timeline.txt | head -50
rm -rf $OUT/s1_prof
