#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_hip.json 2>$O/bench_hip.err
tail -4 $O/pytest_gpu.log; python -c "
import json; r=json.load(open('$O/bench_hip.json')); print('hip', r['value'], r['ms_per_step']); [print('   ',k,v) for k,v in r['kernels'].items()]"
