#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprofv3 kernel stats. Logs -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m rnnpose_amd.build > $O/build.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $O/device.txt
nproc >> $O/device.txt; lscpu | grep "Model name" >> $O/device.txt
if [ "${1:-all}" != "benchonly" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
fi
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
timeout 300 python tools/conv_probe.py $O/conv_probe.json > $O/conv_probe.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/prof.log 2>&1 )
find $O/prof -name "*stats*" | head
tail -5 $O/pytest_gpu.log; cat $O/smoke.log | tail -3; cat $O/bench.json | cut -c1-1500; tail -3 $O/bench.err
