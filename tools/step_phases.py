#!/usr/bin/env python3
"""Phases of one steady-state refinement step from a rocprofv3 --kernel-trace database: per outer iteration, when each encoder image
set runs (first stem .. output convolution), when each loop chain runs (first induced_coords .. last lm), and how much of each phase
has 1 / 2 / 3+ kernels in flight.  (r06: does the two-branch outer graph overlap its encoder sets?)
    python tools/step_phases.py gpurun_out/<tag>_prof/run_results.db"""
import sqlite3
import sys

from timeline import short


def conc(rows, a, b):
    ev = []
    for n, s, e, q in rows:
        s, e = max(s, a), min(e, b)
        if e > s:
            ev += [(s, 1), (e, -1)]
    ev.sort()
    cur, last, h = 0, a, {}
    for t, d in ev:
        h[cur] = h.get(cur, 0) + t - last
        cur += d
        last = t
    h[cur] = h.get(cur, 0) + b - last
    tot = max(1, b - a)
    return " ".join(f"{k}:{v / tot * 100:.0f}%" for k, v in sorted(h.items()))


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = [(short(n), s, e, q) for n, s, e, q in db.execute("select name, start, end, queue_id from kernels order by start")]
    stems = [i for i, r in enumerate(rows) if "stem_conv" in r[0]]
    nst = 6
    i0, i1 = stems[-2 * nst], stems[-nst]
    step = rows[i0:i1]
    t0, t1 = step[0][1], rows[i1][1]
    print(f"step {(t1 - t0) / 1e6:.3f} ms, {len(step)} dispatches; whole step in flight: {conc(step, t0, t1)}")
    us = lambda t: (t - t0) / 1e3
    st = [r for r in step if "stem_conv" in r[0]]
    pyr = [r for r in step if "corr_pyramid" in r[0]]
    ind = [r for r in step if "induced_coords" in r[0]]
    lm = [r for r in step if "lm_normal_eq" in r[0]]
    for o in range(3):
        s_a, s_b = st[2 * o], st[2 * o + 1]
        p = pyr[o]
        print(f"outer {o}: stems at {us(s_a[1]):9.1f} (q{s_a[3]}) and {us(s_b[1]):9.1f} (q{s_b[3]}) us, volume build {us(p[1]):9.1f} .. {us(p[2]):9.1f}; "
              f"encoder phase {us(p[1]) - us(s_a[1]):7.1f} us, in flight: {conc(step, s_a[1], p[1])}")
        first = min(r[1] for r in ind[16 * o:16 * o + 16])
        last = max(r[2] for r in lm[16 * o:16 * o + 16])
        qs = sorted(set(r[3] for r in ind[16 * o:16 * o + 16]))
        per = []
        for q in qs:
            a = min(r[1] for r in ind[16 * o:16 * o + 16] if r[3] == q)
            b = max(r[2] for r in lm[16 * o:16 * o + 16] if r[3] == q)
            per.append(f"q{q}: {us(a):9.1f} .. {us(b):9.1f}")
        print(f"         loop phase {us(first):9.1f} .. {us(last):9.1f} = {us(last) - us(first):7.1f} us ({'; '.join(per)}), in flight: {conc(step, first, last)}")


if __name__ == "__main__":
    main()
