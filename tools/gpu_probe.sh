#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
python -m rnnpose_amd.build > $O/build.log 2>&1
timeout 300 python tools/drift_probe.py $O/drift_default.json > $O/drift_default.log 2>&1
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 300 python tools/drift_probe.py $O/drift_nowino.json > $O/drift_nowino.log 2>&1
MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_DIRECT=0 timeout 300 python tools/drift_probe.py $O/drift_gemmonly.json > $O/drift_gemmonly.log 2>&1
MIOPEN_ENABLE_LOGGING_CMD=1 timeout 200 python tools/drift_probe.py /dev/null 2>&1 | grep -E "MIOpenDriver|Solution|solver" | sort | uniq -c | head -40 > $O/miopen_cmds.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "loop or s2_shape" > $O/pytest_loop.log 2>&1
paste $O/drift_default.log $O/drift_nowino.log | cut -c1-200 | head -60
tail -3 $O/smoke.log; tail -8 $O/pytest_loop.log
