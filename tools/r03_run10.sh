#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python tools/bench_parity_probe.py 2>&1 | grep -v amdgpu.ids | grep -A4 '"flow_up"\|"corr"\|recurrent' | head -40
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > gpurun_out/r03j_pytest_gpu.log 2>&1; tail -25 gpurun_out/r03j_pytest_gpu.log
python bench.py --steps 20 --warmup 3 --cpu-runs 1 > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.err; tail -c 300 gpurun_out/r03j_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03j_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['chip_level']['frac_of_fp16_mfma_peak']); print(d['parity'])
PY
