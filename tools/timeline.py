#!/usr/bin/env python
"""Idle-gap analysis of a rocprofv3 rocpd kernel trace: busy fraction of the steady-state tail and the
kernel pairs between which the device sat idle the longest (summed over occurrences)."""
import sqlite3
import sys
from collections import defaultdict


def main(path, out=None, marker="k_plane_regs", nsteps=4):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else cols[0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    # window = the last `nsteps` steps, delimited by a kernel that runs exactly once per step
    marks = [r[1] for r in rows if r[0].startswith(marker)]
    lo, hi = marks[-nsteps - 1], marks[-1]
    rows = [r for r in rows if lo <= r[1] < hi]
    span = rows[-1][2] - rows[0][1]
    busy = 0
    gaps = defaultdict(lambda: [0, 0])
    end = rows[0][1]
    prev = None
    for n, s, e in rows:
        if s > end:
            if prev is not None:
                g = gaps[(prev, n)]
                g[0] += s - end
                g[1] += 1
            busy += e - s
            end = e
        else:
            if e > end:
                busy += e - end
                end = e
        prev = n
    lines = [f"steps,{nsteps}", f"span_ns,{span}", f"busy_ns,{busy}", f"busy_frac,{busy / span:.4f}", f"kernels,{len(rows)}", "prev,next,total_gap_ns,count,avg_gap_ns"]
    for (a, b), (g, k) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:40]:
        lines.append(f"\"{a[:60]}\",\"{b[:60]}\",{g},{k},{g / k:.0f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], out=sys.argv[2] if len(sys.argv) > 2 else None)
