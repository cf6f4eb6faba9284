R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
{
for n in wait0 tsc1 noslp; do RNNPOSE_LIB=$R/gpurun_extra/det_cw_$n.so timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v "amdgpu.ids\|diag" | head -2; done
for m in c1x1 conv flowfeat lm; do PAIR_B=$m timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v "amdgpu.ids\|diag" | head -2; done
} > $OUT/r05_det_pair6.txt 2>&1
cat $OUT/r05_det_pair6.txt
