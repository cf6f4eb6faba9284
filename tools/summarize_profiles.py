#!/usr/bin/env python3
"""Turns the scratch outputs of tools/gpu_profile.sh (gpurun_out/) into the committed summaries under profiles/:
r01_bench.json, r01_bench_miopen_convs.json, r01_bench_kernel_stats[_nograph].csv (rocprofv3 --kernel-trace --stats),
r01_pmc_traffic_bench.json (FETCH_SIZE / WRITE_SIZE per kernel, FETCH_SIZE doubled: gfx950 correction of
MI355X_MICROARCH.md) and traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic)."""
import collections
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def kname(s):
    s = s.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I", s)
    return m.group(1) if m else re.split(r"[(<]", s)[0]


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(G, d, "k_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            k = kname(r["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


for src, dst in (("bench.json", "r01_bench.json"), ("bench_miopen.json", "r01_bench_miopen_convs.json"),
                 ("prof_stats/bench_kernel_stats.csv", "r01_bench_kernel_stats.csv"),
                 ("prof_stats_nograph/bench_kernel_stats.csv", "r01_bench_kernel_stats_nograph.csv")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
f, w = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    nf, vf = f.get(k, [0, 0])
    nw, vw = w.get(k, [0, 0])
    fetch, write = vf / max(nf, 1) * 1024 * 2, vw / max(nw, 1) * 1024
    out[k] = {"launches": nf, "fetch_MB_per_launch": round(fetch / 1e6, 2), "write_MB_per_launch": round(write / 1e6, 2),
              "hbm_MB_per_launch": round((fetch + write) / 1e6, 2)}
    print(f"{k[:44]:44s} n={nf:5d} fetch {fetch / 1e6:9.1f} MB  write {write / 1e6:9.1f} MB")
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --steps 1 "
                     "--warmup 1 --no-graph`; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md); per-launch means",
           "per_kernel": out}, open(os.path.join(P, "r01_pmc_traffic_bench.json"), "w"), indent=1)
tr = {"source": "profiles/r01_pmc_traffic_bench.json (average over all launches of the kernel in one bench step)"}
for key, kern in (("conv_igemm_bytes_per_launch", "conv_igemm_f16x3_kernel"), ("corr_pyramid_h3_bytes_per_launch", "corr_pyramid_h3_kernel"),
                  ("corr_pyramid_bytes_per_launch", "corr_pyramid_kernel")):
    if kern in out:
        tr[key] = int(out[kern]["hbm_MB_per_launch"] * 1e6)
old = json.load(open(os.path.join(P, "traffic.json"))) if os.path.exists(os.path.join(P, "traffic.json")) else {}
for k, v in old.items():
    tr.setdefault(k, v)
json.dump(tr, open(os.path.join(P, "traffic.json"), "w"))
rows = list(csv.DictReader(open(os.path.join(P, "r01_bench_kernel_stats.csv"))))
print("total kernel ms in the rocprof run:", round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, 1))
for r in rows[:12]:
    print(f'  {kname(r["Name"])[:44]:44s} n={r["Calls"]:>5s} avg={float(r["AverageNs"]) / 1e3:8.1f} us  {float(r["Percentage"]):5.1f} %')
