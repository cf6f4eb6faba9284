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
    # a step = one refinement = 3 outer iterations = 6 stem launches (2 image sets each).  Take the last step that is
    # followed by another one: [first stem of the step, first stem of the next step)
    stems = [i for i, r in enumerate(rows) if "stem_conv" in r[0]]
    nst = int(sys.argv[sys.argv.index("--stems") + 1]) if "--stems" in sys.argv else 6
    if len(stems) < 3 * nst:
        print("not enough steps in the trace")
        return
    i0, i1 = stems[-2 * nst], stems[-nst]
    step = rows[i0:i1]
    t0, t1 = step[0][1], rows[i1][1]
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
        lk = [i for i, r in enumerate(step) if "induced_coords" in r[0]]      # one per inner iteration
        a = lk[len(lk) // 2]
        b = lk[len(lk) // 2 + 1]
        base = step[a][1]
        print(f"--- inner iteration: {(step[b][1] - base) / 1e3:.1f} us per inner iteration")
        for n, s, e, q, st in step[a:b][:N]:
            print(f"  +{(s - base) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  q{q} s{st}  {n}")
        # and the encoder/outer part
        base = step[0][1]
        print(f"--- outer part: first {N} dispatches")
        for n, s, e, q, st in step[:N]:
            print(f"  +{(s - base) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  q{q} s{st}  {n}")


if __name__ == "__main__":
    main()
