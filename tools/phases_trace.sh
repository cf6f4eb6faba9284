R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OUT/ph_prof -o run -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/ph_prof.log 2>&1 )
cd $R/tools && python step_phases.py $OUT/ph_prof/run_results.db > $OUT/r06_step_phases.txt 2>&1
cat $OUT/r06_step_phases.txt
python - <<PY
import sqlite3,sys
sys.path.insert(0,"$R/tools")
from timeline import short
db=sqlite3.connect("$OUT/ph_prof/run_results.db")
rows=[(short(n),s,e,q) for n,s,e,q in db.execute("select name,start,end,queue_id from kernels order by start")]
stems=[i for i,r in enumerate(rows) if "stem_conv" in r[0]]
i0,i1=stems[-12],stems[-6]
with open("$OUT/r06_step_timeline.csv","w") as f:
    t0=rows[i0][1]
    for n,s,e,q in rows[i0:i1]:
        f.write(f"{n[:60]},{q},{(s-t0)/1e3:.1f},{(e-s)/1e3:.1f}\n")
PY
rm -rf $OUT/ph_prof
