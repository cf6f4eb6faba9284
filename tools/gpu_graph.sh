#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "loop or s2_shape or teacher or facade" > $O/pytest_graph.log 2>&1; echo "rc=$?" >> $O/pytest_graph.log
tail -6 $O/pytest_graph.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== S1 240x240 B=1 3x4"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --height 240 --width 240 --batch 1 --inner 4 2>$O/s1.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"; tail -2 $O/s1.err
echo "== S2"; timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>$O/s2.err | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"; tail -2 $O/s2.err
