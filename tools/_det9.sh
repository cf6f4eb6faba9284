R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
{
timeout 300 python tools/visibility_pair_probe.py 300 2>&1 | grep -v "amdgpu.ids\|diag" | head -3
RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so timeout 300 python tools/visibility_pair_probe.py 300 2>&1 | grep -v "amdgpu.ids\|diag" | head -3
PAIR_B=c1x1 RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v "amdgpu.ids\|diag" | head -3
PAIR_ONE_STREAM=1 RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so timeout 300 python tools/visibility_pair_probe.py 100 2>&1 | grep -v "amdgpu.ids\|diag" | head -3
} > $OUT/r05_det_pair_final.txt 2>&1
cat $OUT/r05_det_pair_final.txt
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/r05_b_pytest_gpu.log 2>&1; tail -8 $OUT/r05_b_pytest_gpu.log
{
  echo "== two chains + two encoder streams, in-tree, 20 instances"; DET_ENCODER=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 RNNPOSE_SPLIT_BATCH=1 timeout 600 python tools/determinism_probe.py 20 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== two chains, pointwise.hip with the SLP vectoriser (r04 build), 10 instances"; RNNPOSE_LIB=$R/gpurun_extra/det_pw_slp.so RNNPOSE_SPLIT_BATCH=1 timeout 600 python tools/determinism_probe.py 10 2>&1 | grep -v amdgpu.ids | tail -4
} > $OUT/r05_det_library_final.txt 2>&1
cat $OUT/r05_det_library_final.txt
b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "iters/s", d["ms_per_step"], "ms/step")'; }
{
echo "default: $(b)"
echo "two chains: $(RNNPOSE_SPLIT_BATCH=1 b)"
echo "two chains + two encoder streams: $(RNNPOSE_SPLIT_BATCH=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 b)"
echo "default again: $(b)"
echo "two chains + two encoder streams again: $(RNNPOSE_SPLIT_BATCH=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 b)"
echo "S1: $(b --batch 1 --height 240 --width 240 --inner 4)"
echo "S1 helper stream: $(RNNPOSE_SIDE_STREAM=1 b --batch 1 --height 240 --width 240 --inner 4)"
echo "B16 240: $(b --batch 16 --height 240 --width 240 --inner 4 --steps 10)"
echo "B16 240 two chains: $(RNNPOSE_SPLIT_BATCH=1 b --batch 16 --height 240 --width 240 --inner 4 --steps 10)"
} 2>&1 | tee $OUT/r05_b_schedules.txt
python tools/tail_kernels.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_b_tail_kernels.txt
