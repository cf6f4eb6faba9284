#!/bin/bash
# r05 GPU pass behind profiles/r05_determinism.txt (the packed-fp32 / 16x16x32-MFMA finding): the two stand-alone probes, the library
# pair probe under the in-tree build and under the build that reproduces r04's finding (bash tools/det_variants.sh first: pw_slp =
# pointwise.hip with the SLP vectoriser on), and the whole-loop probe under the two-stream schedules.
#   gpurun --timeout 900 -- 'bash tools/det_round.sh'  -> gpurun_out/r05_det_*.txt      (~3 minutes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 tools/probes/pk_f32_vs_mfma 2000 | tee $OUT/r05_det_pk_probe.txt
timeout 300 tools/probes/kernel_visibility 3000 > $OUT/r05_det_reproducer.txt 2>&1; tail -5 $OUT/r05_det_reproducer.txt
P="timeout 300 python tools/visibility_pair_probe.py"
{
  $P 300
  RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so $P 300
  PAIR_B=c1x1 RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so $P 100
  PAIR_ONE_STREAM=1 RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so $P 100
} 2>&1 | grep -v "amdgpu.ids\|diag" > $OUT/r05_det_pair_final.txt
cat $OUT/r05_det_pair_final.txt
{
  echo "== two chains + two encoder streams, in-tree"; DET_ENCODER=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_SPLIT_BATCH=1 timeout 600 python tools/determinism_probe.py 20 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, pointwise.hip with the SLP vectoriser (r04's build)"; RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so RNNPOSE_SPLIT_BATCH=1 timeout 600 python tools/determinism_probe.py 10 2>&1 | grep -v amdgpu.ids | tail -4
} > $OUT/r05_det_library_final.txt 2>&1
cat $OUT/r05_det_library_final.txt
