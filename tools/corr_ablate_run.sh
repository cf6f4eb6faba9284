#!/bin/bash
# On the GPU box: the correlation-volume build alone under every diagnostics build tools/corr_ablate.sh left in rnnpose_amd/lib
# (cv_<name>.so), then the store-pattern probe.   bash tools/corr_ablate_run.sh <tag>  -> gpurun_out/<tag>_corr_ablation.txt, <tag>_corr_store_patterns.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; T=${1:-r03}; mkdir -p gpurun_out
{
echo "# tools/corr_variants.py: B=8, 60x80, C=256, 4 levels; whole C-ABI call from fp32 pixel-major maps (pre-pass + kernel) | from split operands (kernel only)"
echo "# RPC_ABL bits: 1 K loop requests no operands after the first slab, 2 no epilogue stores, 4 no K loop, 8 no epilogue, 16 K loop does not refill LDS,"
echo "#               32 no level-0 stores, 64 no pooled-level stores;  nt0 / nt2: level-0 stores temporal / all levels non-temporal"
CORR_FILL=1 python tools/corr_variants.py 2>&1 | tail -2
for f in $(ls gpurun_extra/cv_*.so 2>/dev/null | sort -V); do RNNPOSE_LIB=$R/$f python tools/corr_variants.py 2>&1 | tail -1; done
python tools/corr_variants.py 2>&1 | tail -1
echo "# other shapes (in-tree build): B=16 30x30 (LINEMOD crops), B=8 120x160 (960x1280 images)"
python tools/corr_variants.py 16 30 30 2>&1 | tail -1
python tools/corr_variants.py 8 120 160 2>&1 | tail -1
} > gpurun_out/${T}_corr_ablation.txt 2>&1
if [ -x tools/probes/store_pattern ]; then tools/probes/store_pattern > gpurun_out/${T}_corr_store_patterns.txt 2>&1; fi
cat gpurun_out/${T}_corr_ablation.txt
