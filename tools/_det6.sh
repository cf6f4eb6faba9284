R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
{
timeout 300 python tools/visibility_pair_probe.py 60 2>&1 | grep -v amdgpu.ids | head -12
HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v amdgpu.ids | head -3
HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v amdgpu.ids | head -3
} > $OUT/r05_det_pair4.txt 2>&1
cat $OUT/r05_det_pair4.txt
