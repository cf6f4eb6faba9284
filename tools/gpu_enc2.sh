#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m rnnpose_amd.build > gpurun_out/build.log 2>&1
timeout 300 python tools/enc_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/enc_bench.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_hip.json 2>gpurun_out/bench_hip.err
python -c "
import json; r=json.load(open('gpurun_out/bench_hip.json')); print('hip', r['value'], r['ms_per_step']); print(r['roofline']); print(r['correlation_volume_kernel'])"
