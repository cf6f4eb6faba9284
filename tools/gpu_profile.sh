#!/bin/bash
# Full round-end style run: all GPU tests, smoke, bench (+CPU baseline), rocprofv3 kernel stats (csv) and PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o k -- python $R/tools/pmc_kernels.py 3 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o k -- python $R/tools/pmc_kernels.py 3 > $O/pmc_write.log 2>&1
cd $R
find $O/prof_stats $O/pmc_fetch $O/pmc_write -type f | head -30
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-600 $O/bench.json; tail -2 $O/bench.err
