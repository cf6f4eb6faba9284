R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
{
PAIR_CONSUMER=copy timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids | head -8
PAIR_CONSUMER=copy PAIR_B=producer timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids | head -5
PAIR_DELAY=200000 timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v amdgpu.ids | head -5
PAIR_DELAY=2000000 timeout 300 python tools/visibility_pair_probe.py 50 2>&1 | grep -v amdgpu.ids | head -5
PAIR_B=fill timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids | head -5
PAIR_B=copy timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids | head -5
GPU_MAX_HW_QUEUES=1 timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v amdgpu.ids | head -3
} > $OUT/r05_det_pair3.txt 2>&1
cat $OUT/r05_det_pair3.txt
