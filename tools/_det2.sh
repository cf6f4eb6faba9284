R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
P="timeout 400 python tools/determinism_probe.py 10"
{
  echo "== two chains, the COMPLETE r04 library (git archive of e4985d0, built with the same flags)"; RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_r04_lib.so $P 2>&1 | grep -v amdgpu.ids | tail -6
  echo "== two chains, r04 mask_upsample object in the r05 library"; RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_mu_r04.so $P 2>&1 | grep -v amdgpu.ids | tail -6
  echo "== two chains, r04 library, encoder in the loop, two encoder streams"; DET_ENCODER=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 RNNPOSE_SPLIT_BATCH=1 RNNPOSE_LIB=$R/gpurun_extra/det_r04_lib.so $P 2>&1 | grep -v amdgpu.ids | tail -6
  echo "== two chains, in-tree, 16 instances"; RNNPOSE_SPLIT_BATCH=1 timeout 400 python tools/determinism_probe.py 16 2>&1 | grep -v amdgpu.ids | tail -6
} > $OUT/r05_det_library2.txt 2>&1
cat $OUT/r05_det_library2.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/r05_a_bench.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/r05_a_bench.json')); print('headline', d['value'], d['ms_per_step'])"
RNNPOSE_SPLIT_BATCH=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/r05_a_bench_two_chains.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/r05_a_bench_two_chains.json')); print('two chains', d['value'], d['ms_per_step'])"
