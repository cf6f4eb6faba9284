#!/bin/bash
# Full round-end style run: all GPU tests, smoke, bench (+CPU baseline), rocprofv3 kernel stats (csv) and PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --conv-backend miopen > $O/bench_miopen.json 2> $O/bench_miopen.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_nograph -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/prof_stats_nograph.log 2>&1
K='conv_igemm|corr_pyramid|corr_weight|corr_lookup|convex_upsample|lm_normal|instnorm'
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$K" --output-format csv -d $O/pmc_fetch -o k -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$K" --output-format csv -d $O/pmc_write -o k -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_write.log 2>&1
cd $R
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; tail -2 $O/bench.err
