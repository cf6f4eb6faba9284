#!/bin/bash
# rocprofv3 kernel trace of a short bench run reduced to a per-kernel table:  gpurun -- 'bash tools/quick_trace.sh <tag> [ENV=..]'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1 )
python - <<PY
import sys, os
sys.path.insert(0, "$R/tools")
sys.argv = ["x", "$TAG", "$OUT/${TAG}_summary"]
import sqlite3, csv, pmc_report
db = sqlite3.connect("$OUT/${TAG}_prof/run_results.db")
rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open("$OUT/${TAG}_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ms,mean_us,pct\n")
    for n, c, s, a in rows:
        f.write(f"\"{pmc_report.short(n)}\",{c},{s/1e6:.3f},{a/1e3:.2f},{100*s/tot:.2f}\n")
print(open("$OUT/${TAG}_kernel_stats.csv").read()[:3000])
PY
rm -rf $OUT/${TAG}_prof
