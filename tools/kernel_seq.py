#!/usr/bin/env python
"""Print the durations (us) of every launch of one kernel in the last N steps of a rocpd kernel trace, in launch order."""
import sqlite3
import sys


def main(path, kernel="k_wgrad(", nsteps=2, marker="k_plane_regs"):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else cols[0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [r[1] for r in rows if r[0].startswith(marker)]
    lo, hi = marks[-nsteps - 1], marks[-1]
    out = [round((e - s) / 1000.0, 1) for n, s, e in rows if lo <= s < hi and n.startswith(kernel)]
    print(kernel, out)


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), **({"nsteps": int(sys.argv[3])} if len(sys.argv) > 3 else {}))
