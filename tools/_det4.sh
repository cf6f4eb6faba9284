R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
timeout 400 tools/probes/kernel_visibility 3000 "B tiny" > $OUT/r05_det_reproducer_tiny.txt 2>&1
timeout 200 tools/probes/kernel_visibility 3000 "stream B:" >> $OUT/r05_det_reproducer_tiny.txt 2>&1
cat $OUT/r05_det_reproducer_tiny.txt
{
for lib in "" det_wt det_rel det_wt_acq; do
  RNNPOSE_LIB=${lib:+$R/gpurun_extra/$lib.so} timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids | head -4
done
for m in tiny producer consumer; do
  PAIR_B=$m timeout 300 python tools/visibility_pair_probe.py 200 2>&1 | grep -v amdgpu.ids | head -4
done
} > $OUT/r05_det_pair2.txt 2>&1
cat $OUT/r05_det_pair2.txt
