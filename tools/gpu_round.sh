#!/bin/bash
# One GPU-box pass for the round's record: all -m gpu tests, smoke(), bench.py (+ other shapes / schedules), rocprofv3 kernel traces,
# SQ / HBM counter passes.  Run through gpurun:
#   gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r05 [noprof|lean]'
# Everything lands in gpurun_out/<tag>_*; tools/summarize_profiles.py <tag> rNN copies the summaries into profiles/.
# lean (r05: 90 GPU-minutes per round, most of them spent on the reproducibility finding): the counter passes are reduced to HBM
# traffic over bench.py + the two SQ passes over the convolution layers; no drift / error-budget / ablation legs.
TAG=${1:-r05}
MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{ rocminfo 2>/dev/null | grep -E "^Agent|Marketing Name|Compute Unit|Max Clock Freq|Name: +gfx|Device Type|Wavefront Size"     # every agent: the EPYC host and the gfx950 GPU
  python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(f"torch {torch.__version__}, HIP {torch.version.hip}: device 0 = '{p.name}' arch {p.gcnArchName}, {p.multi_processor_count} CUs, {p.total_memory / 2**30:.0f} GiB, "
      f"nominal clock {getattr(p, 'clock_rate', 0) / 1000:.0f} MHz")
PY
  timeout 120 tools/probes/mfma_clock
} > $OUT/${TAG}_device.txt 2>&1
tail -6 $OUT/${TAG}_device.txt
( time python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $OUT/${TAG}_pytest_gpu.log 2>&1
tail -5 $OUT/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
B="python bench.py --no-cpu-baseline"
if [ "$MODE" = "noprof" ]; then
  python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  tail -c 300 $OUT/${TAG}_bench.err; head -c 300 $OUT/${TAG}_bench.json; echo
  exit 0
fi
# HBM traffic of the very launches bench.py times (FETCH_SIZE / WRITE_SIZE: separate passes, --kernel-trace only) -> traffic.json first,
# so that the headline line below carries THIS tree's counters (bench.py refuses a traffic.json measured on other sources)
PMC_TRAFFIC_ONLY=1 bash tools/pmc_sq.sh ${TAG}_b python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
if [ "$MODE" = "lean" ]; then
  PMC_SQ_ONLY=1 bash tools/pmc_sq.sh ${TAG}_c python tools/conv_layers.py 3 > /dev/null 2>&1
  bash tools/pmc_sq.sh ${TAG}_k python tools/pmc_kernels.py 3 > /dev/null 2>&1       # (lm_normal_eq's VALU / LDS utilisation, the lookup's traffic)
else
  bash tools/pmc_sq.sh ${TAG}_k python tools/pmc_kernels.py 3 > /dev/null 2>&1
  bash tools/pmc_sq.sh ${TAG}_c python tools/conv_layers.py 3 > /dev/null 2>&1
fi
python tools/summarize_profiles.py $TAG $OUT/${TAG}_summary > $OUT/${TAG}_summary.log 2>&1
[ -f $OUT/${TAG}_summary/traffic.json ] && cp $OUT/${TAG}_summary/traffic.json profiles/traffic.json
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 300 $OUT/${TAG}_bench.err; head -c 300 $OUT/${TAG}_bench.json; echo
RNNPOSE_SPLIT_BATCH=0 RNNPOSE_ENCODER_MERGE=1 $B --steps 20 --warmup 3 > $OUT/${TAG}_bench_one_stream.json 2>/dev/null     # r04's default: one chain, one stream
$B --steps 20 --warmup 3 --batch 1 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_S1.json 2>/dev/null
$B --steps 10 --warmup 2 --batch 16 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_B16_240.json 2>/dev/null   # configs[2] shape
$B --steps 10 --warmup 2 --batch 32 --height 240 --width 240 --inner 4 > $OUT/${TAG}_bench_B32_240.json 2>/dev/null   # configs[3] shape
$B --steps 5 --warmup 2 --height 960 --width 1280 > $OUT/${TAG}_bench_S5.json 2>/dev/null
$B --steps 20 --warmup 3 --mixed-precision > $OUT/${TAG}_bench_mixed_precision.json 2>/dev/null
if [ "$MODE" != "lean" ]; then
  $B --steps 20 --warmup 3 --no-graph > $OUT/${TAG}_bench_nograph.json 2>/dev/null
  $B --steps 10 --warmup 2 --batch 16 > $OUT/${TAG}_bench_B16.json 2>/dev/null
  RNNPOSE_SPLIT_TENSORS=0 $B --steps 20 --warmup 3 > $OUT/${TAG}_bench_fp32_activations.json 2>/dev/null
  timeout 600 python tools/error_budget.py > $OUT/${TAG}_error_budget.log 2>&1; cp $OUT/error_budget.json $OUT/${TAG}_error_budget.json
  timeout 900 python tools/parity_probe.py > $OUT/${TAG}_parity_probe.log 2>&1; cp $OUT/parity_probe.json $OUT/${TAG}_parity_probe.json
  python tools/drift_probe.py > $OUT/${TAG}_drift.log 2>&1; cp $OUT/drift_probe.json $OUT/${TAG}_drift.json; tail -1 $OUT/${TAG}_drift.log
fi
python tools/conv_layers.py > $OUT/${TAG}_conv_layers_alone.txt 2>&1
python tools/encoder_layers.py > $OUT/${TAG}_encoder_layers.txt 2>&1
python tools/tail_kernels.py > $OUT/${TAG}_tail_kernels.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  # (1) the default schedule as the timed steps run it: hipGraph replay, two loop chains + two encoder streams OVERLAPPING
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
  # (2) the same launches one at a time (AMD_SERIALIZE_KERNEL=3: the runtime waits for every kernel; eager): each launch ALONE on the
  #     chip = the durations bench.py's HIP events measure and roofline.frac is defined on
  AMD_SERIALIZE_KERNEL=3 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_serial -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/${TAG}_prof_serial.log 2>&1
  # (3) r04's default (one chain, one stream), graph replay
  RNNPOSE_SPLIT_BATCH=0 RNNPOSE_ENCODER_MERGE=1 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_one_stream -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof_one_stream.log 2>&1
  rocprofv3 --kernel-trace -d $OUT/${TAG}_prof_convs -o run -- python $R/tools/conv_layers.py 7 > $OUT/${TAG}_prof_convs.log 2>&1 )
# summaries on the box (gpurun merges <= 64 MiB back), then drop the databases
python tools/summarize_profiles.py $TAG $OUT/${TAG}_summary >> $OUT/${TAG}_summary.log 2>&1; tail -4 $OUT/${TAG}_summary.log
for f in encoder_layers tail_kernels; do grep -v amdgpu.ids $OUT/${TAG}_$f.txt > $OUT/${TAG}_summary/${TAG}_$f.txt; done
cp $OUT/${TAG}_bench.json $OUT/${TAG}_summary/${TAG}_bench.json
rm -rf $OUT/${TAG}_*_pmc[0-9] $OUT/${TAG}_prof $OUT/${TAG}_prof_serial $OUT/${TAG}_prof_one_stream $OUT/${TAG}_prof_convs
du -sh $OUT | cut -f1
