#!/bin/bash
# r05 GPU pass for the kernel-to-kernel visibility finding: the stand-alone reproducer, then the library's
# determinism probe under the two-chain schedule with producer- / consumer-side diagnostics builds (bash tools/det_variants.sh),
# system-scope fences (AMD_OPT_FLUSH=0), the single-image shape with its helper stream (eager and graph), two processes on one GPU.
#   gpurun --timeout 1200 -- 'bash tools/det_round.sh'  -> gpurun_out/r05_det_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
N=${DET_TRIALS:-8}
( rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock Freq|Name: +gfx" | sort | uniq -c ) > $OUT/r05_det_device.txt
timeout 300 tools/probes/kernel_visibility 3000 > $OUT/r05_det_reproducer.txt 2>&1
cat $OUT/r05_det_reproducer.txt
P="timeout 400 python tools/determinism_probe.py $N"
{
  echo "== two chains, in-tree library (whole-line stores in mask_upsample)"; RNNPOSE_SPLIT_BATCH=1 $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, mask_upsample with the r04 epilogue"; RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_mu_old.so $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, r04 epilogue, AMD_OPT_FLUSH=0"; AMD_OPT_FLUSH=0 RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_mu_old.so $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, r04 epilogue + acquire in corr_weight"; RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_mu_old_acq.so $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, r04 epilogue + sc1 loads in corr_weight"; RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_mu_old_sc1.so $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, whole-line stores + acquire in corr_weight"; RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_acq.so $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, in-tree, encoder in the loop (two encoder streams)"; DET_ENCODER=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_SPLIT_BATCH=1 $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== one chain (default), in-tree"; $P 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== S1 (B=1 240x240, 4 iterations), helper stream, eager"; DET_B=1 DET_H=240 DET_W=240 DET_ITERS=4 $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== S1, helper stream, graph replay"; DET_GRAPH=1 DET_B=1 DET_H=240 DET_W=240 DET_ITERS=4 $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== S1, helper stream, eager, r04 epilogue"; RNNPOSE_LIB=$R/gpurun_extra/det_mu_old.so DET_B=1 DET_H=240 DET_W=240 DET_ITERS=4 $P 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== S1, no helper stream, eager"; RNNPOSE_SIDE_STREAM=0 DET_B=1 DET_H=240 DET_W=240 DET_ITERS=4 $P 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== two processes on one GPU, one chain each (default schedule)"
  $P > $OUT/r05_det_p1.log 2>&1 &
  P1=$!
  $P > $OUT/r05_det_p2.log 2>&1
  wait $P1
  tail -3 $OUT/r05_det_p1.log $OUT/r05_det_p2.log | grep -v amdgpu.ids
} > $OUT/r05_det_library.txt 2>&1
cat $OUT/r05_det_library.txt
# cost of the choices on this box
for cfg in "" "RNNPOSE_SIDE_STREAM=0"; do
  echo "S1 $cfg: $(env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 --height 240 --width 240 --inner 4 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
done | tee $OUT/r05_det_s1_cost.txt
for lib in "" gpurun_extra/det_mu_old.so; do
  echo "headline lib=${lib:-in-tree}: $(RNNPOSE_LIB=${lib:+$R/$lib} python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "iters/s", d["ms_per_step"], "ms/step", d.get("parity", {}).get("ok"))')"
done | tee $OUT/r05_det_headline.txt
python tools/tail_kernels.py mask_upsample 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/r05_det_mask_upsample.txt
RNNPOSE_LIB=$R/gpurun_extra/det_mu_old.so python tools/tail_kernels.py mask_upsample 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/r05_det_mask_upsample.txt
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "mask_upsample or update_engine or refinement_loop or timed_configuration or two_chains" 2>&1 | tail -5 | tee $OUT/r05_det_pytest.txt
