#!/bin/bash
# Quick perf iteration on the GPU box: the loop-level parity tests, bench.py (no CPU baseline), one rocprofv3 kernel trace.
#   gpurun --timeout 900 -- 'bash tools/gpu_perf.sh <tag> ["pytest -k expression"]'
TAG=${1:-perf}
KEXPR=${2:-"engine or loop or refine or facade or update_block or teacher or short_horizon"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -X faulthandler -m pytest tests -m gpu -v --tb=short -p no:cacheprovider -k "$KEXPR" > $OUT/${TAG}_pytest.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" $OUT/${TAG}_pytest.log | grep -v PASSED | tail -15
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("BENCH", d["value"], "iters/s", d["ms_per_step"], "ms/step chip", d["chip_level"]["frac_of_fp16_mfma_peak"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
ls -la $OUT/${TAG}_prof | tail -2
