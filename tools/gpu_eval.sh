#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m rnnpose_amd.build > gpurun_out/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_eval.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/pytest_eval.log
