#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -s > $O/pytest_conv.log 2>&1; echo "rc=$?" >> $O/pytest_conv.log
timeout 300 python tools/conv_bench.py > $O/conv_bench.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "corr_pyramid" > $O/pytest_pyr.log 2>&1
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_quick.json 2>$O/bench_quick.err
grep -E "split err|passed|failed|Error|error" $O/pytest_conv.log | head -40; cat $O/conv_bench.log; tail -2 $O/pytest_pyr.log; python -c "
import json; r=json.load(open('$O/bench_quick.json')); print(r['value'], r['roofline'])"
