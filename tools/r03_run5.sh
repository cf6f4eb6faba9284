#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > gpurun_out/r03e_pytest_gpu.log 2>&1; tail -40 gpurun_out/r03e_pytest_gpu.log
timeout 900 python tools/parity_probe.py 2>&1 | grep -v amdgpu.ids | tail -45
