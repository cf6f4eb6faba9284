#!/usr/bin/env python3
"""Turns the scratch outputs of tools/gpu_round.sh (gpurun_out/<tag>_*) into the committed record under profiles/:

    python tools/summarize_profiles.py r02 [outdir]   # gpurun_out/r02_* -> profiles/r02_* (or outdir/)

tools/gpu_round.sh runs it ON the GPU box with outdir = gpurun_out/<tag>_summary and then deletes the rocprofv3 databases
(gpurun merges at most 64 MiB back); the summary directory is what gets copied into profiles/.

  <tag>_bench*.json            the bench lines (default config, --no-graph, S1 / S5 / B16 shapes)
  <tag>_kernel_stats*.csv      per-kernel calls / total / mean / min / max of the rocprofv3 --kernel-trace --stats runs of bench.py
                               (graph replay, the default schedule; and the other schedule -- r04: two chains -- with --no-graph)
  <tag>_conv_layers_alone.txt  tools/conv_layers.py: every update-block convolution shape alone on the chip (us, executed TF, % of peak)
  <tag>_conv_layers.csv        the same launches in rocprofv3's kernel trace (keyed by grid size)
  <tag>_pmc_kernels.json       SQ + FETCH/WRITE counters per hand-written kernel (tools/pmc_kernels.py), incl. lm_normal_eq's VALU/LDS
  <tag>_pmc_convs.json         the same per convolution launch shape (tools/conv_layers.py, keyed by grid size)
  <tag>_pmc_traffic_bench.json FETCH/WRITE bytes per kernel over one eager bench step
  <tag>_drift.json             free-running 3x8 drift against the fp64 oracle (tools/drift_probe.py)
  <tag>_conv_ablation.txt      ablation builds of the convolution kernel (tools/conv_ablate.sh): time without weight loads / LDS reads / staging / ...
  <tag>_corr_ablation.txt, <tag>_corr_store_patterns.txt   diagnostics builds of the volume kernel + the store-pattern probe (tools/corr_ablate_run.sh)
  <tag>_error_budget.json      fp64 evaluation vs the fp32 CPU oracle vs the GPU as the feature magnitude grows (tools/error_budget.py)
  <tag>_parity_probe.json      first-iteration flow of the timed configuration: GPU vs both oracle forms, encoder in / out (tools/parity_probe.py)
  <tag>_pytest_gpu.txt, <tag>_smoke.txt, <tag>_device.txt
  traffic.json                 per-launch HBM bytes bench.py reports as roofline.traffic (regenerated here every round)

FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM/rocprofv3 section)."""
import csv
import json
import os
import re
import shutil
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_report  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), (os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "profiles"))


def have(*parts):
    p = os.path.join(G, *parts)
    return p if os.path.exists(p) else None


def kernel_stats(db_path, dst, by_grid=False):
    """rocprofv3's `kernels` view -> csv (name, [grid], calls, total_ms, mean_us, min_us, max_us, pct)."""
    db = sqlite3.connect(db_path)
    grp = "name, grid_x" if by_grid else "name"
    rows = db.execute(f"select {grp}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by {grp} order by 3 desc"
                      if not by_grid else
                      f"select {grp}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by {grp} order by 4 desc").fetchall()
    tot = sum(r[-4] for r in rows) or 1.0
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + (["grid_x"] if by_grid else []) + ["calls", "total_ms", "mean_us", "min_us", "max_us", "pct"])
        for r in rows:
            name = pmc_report.short(r[0])
            extra = [r[1]] if by_grid else []
            n, s, a, lo, hi = r[-5:]
            w.writerow([name] + extra + [n, round(s / 1e6, 3), round(a / 1e3, 2), round(lo / 1e3, 2), round(hi / 1e3, 2), round(100 * s / tot, 2)])
    return tot / 1e6


def tail(src, dst, n=15):
    lines = open(src, errors="replace").read().splitlines()
    open(dst, "w").write("\n".join(lines[-n:]) + "\n")


os.makedirs(P, exist_ok=True)
for suffix in ("bench", "bench_nograph", "bench_S1", "bench_S5", "bench_B16", "bench_B16_240", "bench_B32_240", "bench_split_tensors", "bench_fp32_activations", "bench_two_chains", "bench_one_stream", "bench_mixed_precision", "drift",
               "error_budget", "parity_probe"):
    src = have(f"{TAG}_{suffix}.json")
    if src and os.path.getsize(src):
        shutil.copy(src, os.path.join(P, f"{TAG}_{suffix}.json"))
for suffix, n in (("pytest_gpu.log", 12), ("smoke.log", 4), ("device.txt", 40), ("conv_layers_alone.txt", 40), ("drift.log", 60),
                  ("conv_ablation.txt", 200), ("corr_ablation.txt", 60), ("corr_store_patterns.txt", 40)):
    src = have(f"{TAG}_{suffix}")
    if src:
        tail(src, os.path.join(P, f"{TAG}_{suffix.replace('.log', '.txt')}"), n)
for d, name, by_grid in ((f"{TAG}_prof", f"{TAG}_kernel_stats.csv", False), (f"{TAG}_prof_unsplit", f"{TAG}_kernel_stats_unsplit_nograph.csv", False),
                         (f"{TAG}_prof_serial", f"{TAG}_kernel_stats_serialized.csv", False), (f"{TAG}_prof_one_stream", f"{TAG}_kernel_stats_one_stream.csv", False),
                         (f"{TAG}_prof_convs", f"{TAG}_conv_layers.csv", True)):
    src = have(d, "run_results.db")
    if src:
        print(f"{name}: {kernel_stats(src, os.path.join(P, name), by_grid):.1f} ms of kernel time in the run")

traffic = {}


def pmc(prefix, dst, by_grid=False):
    dbs = [p for p in (have(f"{TAG}_{prefix}_pmc{i}", "run_results.db") for i in range(1, 5)) if p]
    if not dbs:
        return {}
    out = pmc_report.collect(dbs, by_grid, verbose=False)
    for v in out.values():
        if "FETCH_MB" in v or "WRITE_MB" in v:
            v["HBM_MB"] = round(v.get("FETCH_MB", 0.0) + v.get("WRITE_MB", 0.0), 2)
    json.dump({"source": f"tools/pmc_sq.sh over tools/{'conv_layers' if prefix == 'c' else 'pmc_kernels' if prefix == 'k' else 'bench'}.py: "
                         "rocprofv3 --pmc, one pass per counter set, --kernel-trace only; per-launch means; FETCH doubled (gfx950)",
               "per_kernel": out}, open(os.path.join(P, dst), "w"), indent=1)
    print(dst, len(out), "kernels")
    return out


kern = pmc("k", f"{TAG}_pmc_kernels.json")
pmc("c", f"{TAG}_pmc_convs.json", by_grid=True)
bench = pmc("b", f"{TAG}_pmc_traffic_bench.json")
src_of = {}
for key, pat in (("conv", r"^conv_(igemm|strip)_f16x3_kernel"), ("corr_pyramid_h3", r"corr_pyramid_h3_kernel"), ("corr_pyramid", r"corr_pyramid_kernel")):
    for table, label in ((bench, f"{TAG}_pmc_traffic_bench.json"), (kern, f"{TAG}_pmc_kernels.json")):
        rows = [v for k, v in table.items() if re.search(pat, k) and "HBM_MB" in v]
        if rows:
            n = sum(v["launches"] for v in rows)
            traffic[key] = {"bytes_per_launch": int(sum(v["HBM_MB"] * v["launches"] for v in rows) / n * 1e6), "launches": n,
                            "source": f"profiles/{label} (FETCH_SIZE x2 + WRITE_SIZE, mean over the launches of the kernel)"}
            break
if traffic:
    # the counters describe THESE kernel sources: bench.py refuses the file when the digest differs from the tree it runs on
    sys.path.insert(0, ROOT)
    from rnnpose_amd import build as _build
    traffic["csrc_digest"] = _build.source_digest()
    traffic["round"] = TAG
    tp = os.path.join(P, "traffic.json")
    json.dump(traffic, open(tp, "w"), indent=1)
    print("traffic.json:", {k: v["bytes_per_launch"] for k, v in traffic.items() if isinstance(v, dict)})
