#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_conv.log 2>&1; echo "rc=$?" >> $O/pytest_conv.log
timeout 300 python tools/conv_bench.py > $O/conv_bench.log 2>&1
tail -3 $O/pytest_conv.log; cat $O/conv_bench.log
