#!/bin/bash
# One GPU-box pass for the round's record: all -m gpu tests, smoke(), bench.py (+ two other shapes), rocprofv3 kernel traces,
# SQ / HBM counter passes, the drift probe.  Run through gpurun:
#   gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r02 [noprof]'
# Everything lands in gpurun_out/<tag>_*; tools/summarize_profiles.py <tag> rNN copies the summaries into profiles/.
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
tail -c 300 $OUT/${TAG}_bench.err; head -c 300 $OUT/${TAG}_bench.json; echo
if [ "$2" != "noprof" ]; then
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $OUT/${TAG}_bench_nograph.json 2>/dev/null
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_S1.json 2>/dev/null
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --height 960 --width 1280 > $OUT/${TAG}_bench_S5.json 2>/dev/null
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch 16 > $OUT/${TAG}_bench_B16.json 2>/dev/null
  python tools/conv_layers.py > $OUT/${TAG}_conv_layers_alone.txt 2>&1
  python tools/drift_probe.py > $OUT/${TAG}_drift.log 2>&1; cp $OUT/drift_probe.json $OUT/${TAG}_drift.json; tail -1 $OUT/${TAG}_drift.log
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
    rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_unsplit -o run -- env RNNPOSE_SPLIT_BATCH=0 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/${TAG}_prof_unsplit.log 2>&1
    rocprofv3 --kernel-trace -d $OUT/${TAG}_prof_convs -o run -- python $R/tools/conv_layers.py 7 > $OUT/${TAG}_prof_convs.log 2>&1 )
  bash tools/pmc_sq.sh ${TAG}_k python tools/pmc_kernels.py 3 > /dev/null 2>&1
  bash tools/pmc_sq.sh ${TAG}_c python tools/conv_layers.py 3 > /dev/null 2>&1
  PMC_TRAFFIC_ONLY=1 bash tools/pmc_sq.sh ${TAG}_b python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
  # summaries on the box (gpurun merges <= 64 MiB back), then drop the databases except the main kernel trace
  python tools/summarize_profiles.py $TAG $OUT/${TAG}_summary > $OUT/${TAG}_summary.log 2>&1; tail -3 $OUT/${TAG}_summary.log
  rm -rf $OUT/${TAG}_*_pmc[0-9] $OUT/${TAG}_prof_unsplit $OUT/${TAG}_prof_convs
  du -sh $OUT | cut -f1
fi
