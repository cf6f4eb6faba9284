#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CORR_FILL=1 python tools/corr_variants.py 2>&1 | tail -2
for v in "$@"; do RNNPOSE_LIB=$R/rnnpose_amd/lib/corr_$v.so python tools/corr_variants.py 2>&1 | tail -1; done
CORR_FILL=1 python tools/corr_variants.py 2>&1 | tail -2
