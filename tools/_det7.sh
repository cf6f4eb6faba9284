R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
{
for n in 1 2 3; do RNNPOSE_LIB=$R/gpurun_extra/det_cw_dbg$n.so timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v amdgpu.ids | head -9; done
} > $OUT/r05_det_pair5.txt 2>&1
cat $OUT/r05_det_pair5.txt
