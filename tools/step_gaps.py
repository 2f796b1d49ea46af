#!/usr/bin/env python
"""Device idle gaps of one steady-state step of a kernel trace (rocpd sqlite), one stream or several: the step is delimited by a kernel
that runs once per step (default: the PDE prefilter).  usage: step_gaps.py <db> [marker-prefix] [steps-from-end]"""
import sqlite3
import sys


def main(path, marker="void k_rk2_split<", back=3):
    c = sqlite3.connect(path).cursor()
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [r[1] for r in rows if r[0].startswith(marker)]
    lo, hi = marks[-back - 1], marks[-back]
    rows = [r for r in rows if lo <= r[1] < hi]
    busy_end = rows[0][1]
    idle = []
    busy = 0
    cur_s, cur_e = rows[0][1], rows[0][2]
    prev = rows[0][0]
    for n, s, e in rows[1:]:
        if s > cur_e:
            idle.append((s - cur_e, prev, n, cur_e - lo))
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    print(f"step {(hi - lo) / 1e3:.0f} us, {len(rows)} kernels, device busy {busy / 1e3:.0f} us ({busy / (hi - lo):.3f})")
    idle.sort(reverse=True)
    for g, a, b, at in idle[:25]:
        print(f"  idle {g / 1e3:7.1f} us at +{at / 1e3:7.0f} us   {a[:48]:48s} -> {b[:48]}")
    small = sum(g for g, *_ in idle if g < 20000)
    print(f"  gaps < 20 us: {sum(1 for g, *_ in idle if g < 20000)} totalling {small / 1e3:.0f} us; gaps >= 20 us: {sum(1 for g, *_ in idle if g >= 20000)} totalling {sum(g for g, *_ in idle if g >= 20000) / 1e3:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), **({"back": int(sys.argv[3])} if len(sys.argv) > 3 else {}))
