#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider --tb=short -k "alternate or lookup" 2>&1 | tail -6
python tools/corr_alt_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_corr_alt.txt
