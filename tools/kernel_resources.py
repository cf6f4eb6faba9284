#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed):
    python tools/kernel_resources.py conv_igemm.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnpose_amd import build
src = os.path.join(build.CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [build.hipcc(), "-c", "-x", "hip", src, "-o", "/dev/null", "-I", os.path.join(ROOT, "include"), "-I", build.CSRC] + \
      [f for f in build.FLAGS if f != "-shared"] + build.PER_FILE_FLAGS.get(os.path.basename(src), []) + \
      ["-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w \[\]/]*): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs', 0):4d} AGPR {v.get('AGPRs', 0):3d} spill {v.get('VGPRs Spill', 0):3d} "
              f"scratch {v.get('ScratchSize [bytes/lane]', 0):4d} occ {v.get('Occupancy [waves/SIMD]', 0)} LDS {v.get('LDS Size [bytes/block]', 0)}")
