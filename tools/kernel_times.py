#!/usr/bin/env python3
"""Average GPU duration per (kernel, grid) from a rocprofv3 --kernel-trace database.
    python tools/kernel_times.py gpurun_out/<dir>/run_results.db [substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for name, gx, gy, gz, dur in db.execute("select name, grid_x, grid_y, grid_z, duration from kernels order by start"):
    if sub not in name:
        continue
    m = re.search(r"(conv_igemm_f16x3_kernel<[^>]*>|[A-Za-z0-9_]+_kernel)", name)
    key = ((m.group(1) if m else name[:40]), gx, gy, gz)
    a = acc.setdefault(key, [])
    a.append(dur)
for (k, gx, gy, gz), v in acc.items():
    v = sorted(v)
    print(f"{k:52s} grid=({gx},{gy},{gz}) n={len(v):4d} median {v[len(v)//2]/1e3:8.1f} us  min {v[0]/1e3:8.1f} us")
