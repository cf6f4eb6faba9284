#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -x -q -p no:cacheprovider --tb=short -k "patch_tiling or tile_stats" 2>&1 | tail -4
echo "== torchrun, 2 ranks on one GPU (gloo)"; RNNPOSE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 2>gpurun_out/r03p_mr.err | cut -c1-400; tail -3 gpurun_out/r03p_mr.err
echo "== self-spawned, 2 ranks on one GPU (gloo)"; RNNPOSE_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 2>gpurun_out/r03p_mr2.err | cut -c1-400; tail -3 gpurun_out/r03p_mr2.err
ab() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], 'iters/s', d['ms_per_step'], 'ms')"; }
for i in 1 2; do ab RNNPOSE_RANGE_GUARD=1; ab RNNPOSE_RANGE_GUARD=0; ab RNNPOSE_SPLIT_TENSORS=1; done
