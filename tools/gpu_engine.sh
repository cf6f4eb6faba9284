#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "nhwc or engine or teacher or facade or loop_golden or conv" > $O/pytest_engine.log 2>&1; echo "rc=$?" >> $O/pytest_engine.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_hip.json 2>$O/bench_hip.err
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --conv-backend miopen > $O/bench_miopen.json 2>$O/bench_miopen.err
tail -15 $O/pytest_engine.log; for f in hip miopen; do python -c "
import json; r=json.load(open('$O/bench_$f.json')); print('$f', r['value'], r['ms_per_step']); [print('   ',k,v) for k,v in r['kernels'].items()]"; tail -2 $O/bench_$f.err; done
