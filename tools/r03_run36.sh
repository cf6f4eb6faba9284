#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider --tb=short -k "corr_pyramid or split_operands" 2>&1 | tail -3
for st in 8 12 16 20 24 32 40; do CORR_SUPERTILE=$st python tools/corr_variants.py 2>&1 | tail -1; done
for st in 8 20; do CORR_SUPERTILE=$st python tools/corr_variants.py 16 30 30 2>&1 | tail -1; CORR_SUPERTILE=$st python tools/corr_variants.py 8 120 160 2>&1 | tail -1; done
