#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m rnnpose_amd.build > gpurun_out/build.log 2>&1
timeout 600 python tools/conv_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_ablate.log
timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_bench.log
