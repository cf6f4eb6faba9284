#!/bin/bash
# rocprofv3 kernel durations of the stem (and the other encoder kernels) for the in-tree library and variant builds:
#   gpurun -- 'bash tools/stem_ab.sh gpurun_extra/stem_v1.so'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
for lib in "" "$@"; do
  for shape in "8 480 640"; do
    rm -rf /tmp/stem_prof
    ( cd /tmp && export TMPDIR=/tmp && RNNPOSE_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace --stats -d /tmp/stem_prof -o run -- python $R/tools/encoder_layers.py $shape > /dev/null 2>&1 )
    python - <<PY
import sqlite3
db = sqlite3.connect("/tmp/stem_prof/run_results.db")
for n, c, a, mn in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%stem_conv%' group by name"):
    print("[${lib:-in-tree}] ($shape)", n.split("(")[0][-40:], c, "launches, mean %.1f us, min %.1f us" % (a / 1e3, mn / 1e3))
PY
  done
done | tee $OUT/stem_ab.log
