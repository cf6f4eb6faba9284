#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "conv or engine or encoder or eval" > $O/pytest_conv.log 2>&1; echo "rc=$?" >> $O/pytest_conv.log
timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids > $O/conv_bench.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_hip.json 2>$O/bench_hip.err
tail -5 $O/pytest_conv.log; cat $O/conv_bench.log; python -c "
import json; r=json.load(open('$O/bench_hip.json')); print('hip', r['value'], r['ms_per_step'])"
