#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "encoder or instnorm or conv_ or nhwc" > $O/pytest_enc.log 2>&1; echo "rc=$?" >> $O/pytest_enc.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_hip.json 2>$O/bench_hip.err
tail -12 $O/pytest_enc.log; python -c "
import json; r=json.load(open('$O/bench_hip.json')); print('hip', r['value'], r['ms_per_step'])"; tail -2 $O/bench_hip.err
