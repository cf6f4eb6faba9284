#!/bin/bash
# One GPU-box pass: all -m gpu tests, smoke(), bench.py, rocprofv3 kernel stats + trace.  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a'
# Everything lands in gpurun_out/<tag>_*; tools/summarize_profiles.py copies the summaries into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|Compute Unit|gfx" > $OUT/${TAG}_device.txt
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/${TAG}_pytest_gpu.log 2>&1
tail -5 $OUT/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.err; head -c 400 $OUT/${TAG}_bench.json; echo
if [ "$2" != "noprof" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_nograph -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/${TAG}_prof_nograph.log 2>&1
  find $OUT/${TAG}_prof $OUT/${TAG}_prof_nograph -name "*.csv" | head; du -sh $OUT/${TAG}_prof $OUT/${TAG}_prof_nograph
fi
