#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refiner.py tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short -k "corr or encoder or split or lookup or range" 2>&1 | tail -8
python tools/corr_variants.py 2>&1 | tail -1
for v in occ2 occ4 nt0; do RNNPOSE_LIB=$R/rnnpose_amd/lib/corr_$v.so python tools/corr_variants.py 2>&1 | tail -1; done
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['correlation_volume_kernel']; print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms | corr', c['mean_ms'], 'ms frac', c['frac'], '| parity', d['parity']['ok'])"; }

