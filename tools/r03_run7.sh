#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > gpurun_out/r03g_pytest_gpu.log 2>&1; tail -25 gpurun_out/r03g_pytest_gpu.log
python - <<'PY'
import torch, time
from rnnpose_amd import ops
B,H,W=4,480,640
tgt=torch.randn(B,2,H,W,device='cuda'); w=torch.rand(B,H,W,device='cuda'); d=torch.rand(B,1,H,W,device='cuda')+0.9
K=torch.tensor([[572.,0,320],[0,573.,240],[0,0,1]],device='cuda').repeat(B,1,1); G=torch.eye(4,device='cuda').repeat(B,1,1)
for fused in (True, False, True, False):
    ops.lm_fused_tail(fused)
    for _ in range(5): ops.lm_step(tgt,w,d,K,G)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(100): ops.lm_step(tgt,w,d,K,G)
    torch.cuda.synchronize(); print('lm_step fused' if fused else 'lm_step 3 launches', (time.perf_counter()-t0)/100*1e6, 'us')
ops.lm_fused_tail(True)
PY
python bench.py --steps 20 --warmup 3 > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err; tail -c 400 gpurun_out/r03g_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['chip_level']['frac_of_fp16_mfma_peak']); print(d['parity']); print(d['cpu_baseline']['value'], d['cpu_baseline']['min'], d['cpu_baseline']['max'], d['f16x3_range_events'])
PY
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 --height 240 --width 240 --inner 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S1', d['value'], d['ms_per_step'])"
