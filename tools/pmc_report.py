#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_sq.sh (rocprofv3 databases).
    python tools/pmc_report.py gpurun_out/<tag>_pmc1/run_results.db [more dbs ...] [--by-grid] [--json out.json]
FETCH_SIZE is reported doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md) and in MB."""
import json
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"(conv_igemm_f16x3_kernel<[^>]*>|conv_strip_f16x3_kernel<[^>]*>|corr_pyramid_h3_kernel|[A-Za-z0-9_]+_kernel)", n)
    return m.group(1) if m else n[:48]


def collect(dbs, by_grid=False, verbose=True):
    """-> {kernel: {launches, counter means, derived ratios}} over the given rocprofv3 databases."""
    acc = {}
    for path in dbs:
        db = sqlite3.connect(path)
        cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
        q = "select kernel_name, grid_size_x, counter_name, value, dispatch_id from counters_collection" if "grid_size_x" in cols else None
        if q is None:
            gx = "grid_x" if "grid_x" in cols else "grid_size"
            q = f"select kernel_name, {gx}, counter_name, value, dispatch_id from counters_collection"
        per_dispatch = {}
        for name, grid, cname, val, did in db.execute(q):
            key = (short(name), int(grid) if by_grid else 0)
            per_dispatch.setdefault((key, did, cname), 0.0)
            per_dispatch[(key, did, cname)] += float(val)        # (a counter is reported per XCD / SE instance: sum them)
        for (key, did, cname), v in per_dispatch.items():
            a = acc.setdefault(key, {}).setdefault(cname, [0, 0.0])
            a[0] += 1
            a[1] += v
    out = {}
    for key in sorted(acc):
        name = key[0] + (f" grid={key[1]}" if by_grid else "")
        row = {c: v[1] / v[0] for c, v in acc[key].items()}
        n = max(v[0] for v in acc[key].values())
        if "FETCH_SIZE" in row:
            row["FETCH_MB"] = row.pop("FETCH_SIZE") * 1024 * 2 / 1e6
        if "WRITE_SIZE" in row:
            row["WRITE_MB"] = row.pop("WRITE_SIZE") * 1024 / 1e6
        wc = row.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if c in row:
                    row[c + "/WAVE_CYCLES"] = row[c] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "SQ_BUSY_CYCLES" in row and row["SQ_BUSY_CYCLES"]:
            row["MFMA_BUSY/BUSY_CYCLES(raw)"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / row["SQ_BUSY_CYCLES"]   # NOT a utilisation: the two counters are summed over different numbers of instances (divide by ~44 on this part)
        out[name] = dict(launches=n, **{k: (round(v, 4) if v < 100 else round(v, 1)) for k, v in row.items()})
        if verbose:
            print(f"{name[:70]:70s} n={n}")
            for k, v in out[name].items():
                if k != "launches":
                    print(f"      {k:34s} {v}")
    return out


def main():
    out = collect([a for a in sys.argv[1:] if a.endswith(".db")], "--by-grid" in sys.argv)
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
