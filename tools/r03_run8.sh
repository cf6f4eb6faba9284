#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv.py -q --tb=short -p no:cacheprovider -x -k "stem or instnorm or encoder or tile_stats or per_image or timed_configuration" 2>&1 | tail -15
python bench.py --steps 20 --warmup 3 --cpu-runs 1 > gpurun_out/r03h_bench.json 2> gpurun_out/r03h_bench.err; tail -c 300 gpurun_out/r03h_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03h_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['chip_level']['frac_of_fp16_mfma_peak']); print(d['parity'])
k=d['kernels']; print({n: k[n]['mean_ms'] for n in ('stem_conv7x7_s2_f16x3','instnorm_tiles_nhwc_f32','lm_step_io_f32') if n in k})
PY
