R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
timeout 300 tools/probes/kernel_visibility 3000 uniform > $OUT/r05_det_reproducer_uniform.txt 2>&1; cat $OUT/r05_det_reproducer_uniform.txt
{
for lib in det_r04_lib det_mu_r04 det_mu_old ""; do
  RNNPOSE_LIB=${lib:+$R/gpurun_extra/$lib.so} timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids
done
PAIR_ONE_STREAM=1 RNNPOSE_LIB=$R/gpurun_extra/det_r04_lib.so timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids
RNNPOSE_LIB=$R/gpurun_extra/det_mu_old_acq.so timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids
RNNPOSE_LIB=$R/gpurun_extra/det_mu_old_sc1.so timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids
AMD_OPT_FLUSH=0 RNNPOSE_LIB=$R/gpurun_extra/det_r04_lib.so timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids
} > $OUT/r05_det_pair.txt 2>&1
cat $OUT/r05_det_pair.txt
