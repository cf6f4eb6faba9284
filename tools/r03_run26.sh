#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_refiner.py -x -q -p no:cacheprovider --tb=short -k "corr or encoder or split or lookup" 2>&1 | tail -5
python tools/corr_variants.py 2>&1 | tail -1
for v in xpf0 occ2 occ4 abl4 abl2; do RNNPOSE_LIB=$R/rnnpose_amd/lib/corr_$v.so python tools/corr_variants.py 2>&1 | tail -1; done
python tools/corr_variants.py 16 30 30 2>&1 | tail -1
python tools/corr_variants.py 8 120 160 2>&1 | tail -1
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['correlation_volume_kernel']; print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms | corr', c['mean_ms'], 'ms frac', c['frac'])"; }
for i in 1 2; do ab RNNPOSE_SPLIT_FMAPS=1; done
