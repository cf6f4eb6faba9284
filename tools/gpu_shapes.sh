#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
echo "== S5 960x1280 B=8 3x8"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --height 960 --width 1280 2>$O/s5.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['correlation_volume_kernel'])"; tail -2 $O/s5.err
echo "== S1 240x240 B=1 3x4"; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --height 240 --width 240 --batch 1 --inner 4 2>$O/s1.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"; tail -2 $O/s1.err
echo "== torchrun nproc=1"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/tr.err | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['n_gpus'])"; tail -3 $O/tr.err
echo "== B=16 480x640"; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16 2>$O/b16.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['image_iters_per_sec'])"
