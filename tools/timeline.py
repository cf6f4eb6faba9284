#!/usr/bin/env python3
"""Timeline of one steady-state refinement step from a rocprofv3 --kernel-trace database (run_results.db):
which kernels overlap on the two streams, where the chip idles, per-kernel totals inside the step.

    python tools/timeline.py gpurun_out/<tag>_prof/run_results.db [--dump N]
"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"(conv_igemm_f16x3_kernel<[^>]*>|[A-Za-z0-9_]+_kernel|__amd_rocclr_\w+|vectorized_elementwise_kernel|elementwise_kernel\w*|\w+Functor\w*)", n)
    s = m.group(1) if m else n[:40]
    if "FillFunctor" in n:
        s = "aten_fill"
    if "copyBuffer" in n:
        s = "copyBuffer"
    return s


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    rows = [(short(n), s, e, q, st) for n, s, e, q, st in rows]
    # steps are delimited by corr_pyramid launches: 3 per step.  Take the span of the last complete step.
    pyr = [i for i, r in enumerate(rows) if "corr_pyramid_h3" in r[0]]
    if len(pyr) < 7:
        print("not enough steps in the trace")
        return
    # a step starts at the first encoder kernel (stem) before its first volume build
    stems = [i for i, r in enumerate(rows) if "stem_conv" in r[0]]
    # last step = last 3 volume builds
    first_pyr = pyr[-3]
    i0 = max(i for i in stems if i < first_pyr and rows[i][1] < rows[first_pyr][1] and (first_pyr - i) < 400)
    # walk back to the first stem of that encoder pass
    while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 20000 and (first_pyr - i0) < 400:
        i0 -= 1
    step = rows[i0:]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} dispatches")
    # busy union / concurrency histogram
    ev = []
    for n, s, e, q, st in step:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    cur, last, hist = 0, t0, {}
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    tot = t1 - t0
    print("concurrency (kernels in flight): " + ", ".join(f"{k}: {v / tot * 100:.1f}%" for k, v in sorted(hist.items())))
    agg = {}
    for n, s, e, q, st in step:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    print(f"{'kernel':50s} {'calls':>6s} {'sum ms':>9s} {'avg us':>8s}")
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:50s} {c:6d} {d / 1e6:9.3f} {d / c / 1e3:8.1f}")
    if "--dump" in sys.argv:
        N = int(sys.argv[sys.argv.index("--dump") + 1])
        # dump one inner iteration in the middle: from a corr_lookup to the next
        lk = [i for i, r in enumerate(step) if "corr_lookup" in r[0]]
        a = lk[len(lk) // 2]
        b = lk[len(lk) // 2 + 1]
        base = step[a][1]
        print(f"--- inner iteration: {(step[b][1] - base) / 1e3:.1f} us between lookups")
        for n, s, e, q, st in step[a:b][:N]:
            print(f"  +{(s - base) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  q{q} s{st}  {n}")
        # and the encoder/outer part
        base = step[0][1]
        print(f"--- outer part: first {N} dispatches")
        for n, s, e, q, st in step[:N]:
            print(f"  +{(s - base) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  q{q} s{st}  {n}")


if __name__ == "__main__":
    main()
