#!/bin/bash
# One GPU-box pass for the round's record: all -m gpu tests, smoke(), bench.py (+ two other shapes), rocprofv3 kernel traces,
# SQ / HBM counter passes, the drift probe.  Run through gpurun:
#   gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r02 [noprof]'
# Everything lands in gpurun_out/<tag>_*; tools/summarize_profiles.py <tag> rNN copies the summaries into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
rocminfo 2>/dev/null | grep -E "^Agent|Marketing Name|Compute Unit|Max Clock Freq|Name: +gfx|Device Type|Wavefront Size" > $OUT/${TAG}_device.txt   # every agent: the EPYC host and the gfx950 GPU
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/${TAG}_pytest_gpu.log 2>&1
tail -5 $OUT/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 300 $OUT/${TAG}_bench.err; head -c 300 $OUT/${TAG}_bench.json; echo
if [ "$2" != "noprof" ]; then
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $OUT/${TAG}_bench_nograph.json 2>/dev/null
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 1 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_S1.json 2>/dev/null
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --height 960 --width 1280 > $OUT/${TAG}_bench_S5.json 2>/dev/null
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch 16 > $OUT/${TAG}_bench_B16.json 2>/dev/null
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch 16 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_B16_240.json 2>/dev/null   # configs[2] shape
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch 32 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_B32_240.json 2>/dev/null   # configs[3] shape
  RNNPOSE_SPLIT_TENSORS=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_fp32_activations.json 2>/dev/null
  RNNPOSE_SPLIT_BATCH=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_two_chains.json 2>/dev/null   # the r02-r03 schedule (opt-in since r04)
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mixed-precision > $OUT/${TAG}_bench_mixed_precision.json 2>/dev/null
  timeout 600 python tools/error_budget.py > $OUT/${TAG}_error_budget.log 2>&1; cp $OUT/error_budget.json $OUT/${TAG}_error_budget.json
  timeout 900 python tools/parity_probe.py > $OUT/${TAG}_parity_probe.log 2>&1; cp $OUT/parity_probe.json $OUT/${TAG}_parity_probe.json
  if ls gpurun_extra/abl_*.so > /dev/null 2>&1; then     # ablation builds (bash tools/conv_ablate.sh 1 2 4 8 16 32 7 31 in the build container)
    for lib in rnnpose_amd/lib/librnnpose_hip $(ls gpurun_extra | grep -E '^abl_[0-9]+\.so$' | sed 's/\.so//' | sort -t_ -k2 -n | sed 's#^#gpurun_extra/#'); do
      echo "== $lib  (RP_ABL bits: 1 no weight loads, 2 no LDS fragment reads, 4 no activation staging, 8 no barrier, 16 no epilogue stores, 32 half the waves request weights)"
      CONV_LAYERS_FILTER="zr 1x5,q 1x5,heads,convc2,enc l1" CONV_LAYERS_B=4,8,1 RNNPOSE_LIB=$R/$lib.so timeout 200 python tools/conv_layers.py 0 f32,hl1 2>&1 | grep -v amdgpu.ids
    done > $OUT/${TAG}_conv_ablation.txt
  fi
  if ls gpurun_extra/cv_*.so > /dev/null 2>&1; then      # diagnostics builds of the volume kernel (bash tools/corr_ablate.sh ...) + the store-pattern probe
    bash tools/corr_ablate_run.sh $TAG > /dev/null 2>&1
  fi
  python tools/conv_layers.py > $OUT/${TAG}_conv_layers_alone.txt 2>&1
  python tools/drift_probe.py > $OUT/${TAG}_drift.log 2>&1; cp $OUT/drift_probe.json $OUT/${TAG}_drift.json; tail -1 $OUT/${TAG}_drift.log
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
    rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_unsplit -o run -- env RNNPOSE_SPLIT_BATCH=1 RNNPOSE_ENCODER_MERGE=0 RNNPOSE_ENCODER_PARTS=2 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/${TAG}_prof_unsplit.log 2>&1
    rocprofv3 --kernel-trace -d $OUT/${TAG}_prof_convs -o run -- python $R/tools/conv_layers.py 7 > $OUT/${TAG}_prof_convs.log 2>&1 )
  bash tools/pmc_sq.sh ${TAG}_k python tools/pmc_kernels.py 3 > /dev/null 2>&1
  bash tools/pmc_sq.sh ${TAG}_c python tools/conv_layers.py 3 > /dev/null 2>&1
  PMC_TRAFFIC_ONLY=1 bash tools/pmc_sq.sh ${TAG}_b python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
  # summaries on the box (gpurun merges <= 64 MiB back), then drop the databases except the main kernel trace
  python tools/summarize_profiles.py $TAG $OUT/${TAG}_summary > $OUT/${TAG}_summary.log 2>&1; tail -3 $OUT/${TAG}_summary.log
  # the headline line once more, now with THIS pass's counter digest (bench.py refuses a traffic.json measured on other sources)
  cp $OUT/${TAG}_summary/traffic.json profiles/traffic.json
  mv $OUT/${TAG}_bench.json $OUT/${TAG}_bench_first.json
  python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  cp $OUT/${TAG}_bench.json $OUT/${TAG}_summary/${TAG}_bench.json; head -c 200 $OUT/${TAG}_bench.json; echo
  rm -rf $OUT/${TAG}_*_pmc[0-9] $OUT/${TAG}_prof_unsplit $OUT/${TAG}_prof_convs
  du -sh $OUT | cut -f1
fi
