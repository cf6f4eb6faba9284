#!/bin/bash
# rocprofv3 kernel traces of bench.py under two builds of the library, on one box:  bash tools/ab_trace.sh <tag> <lib or ""> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
i=0
for lib in "$@"; do
  i=$((i+1))
  RNNPOSE_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace -d $OUT/${TAG}_$i -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_$i.log 2>&1
  python $R/tools/timeline.py $OUT/${TAG}_$i/run_results.db > $OUT/${TAG}_$i.txt 2>&1
  rm -rf $OUT/${TAG}_$i
  head -12 $OUT/${TAG}_$i.txt
done
