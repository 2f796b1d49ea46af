#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, steps=None, out=None):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else cols[0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, k, s, a, mn, mx in rows:
        lines.append(f"\"{n}\",{k},{s},{a:.1f},{mn},{mx},{100.0 * s / tot:.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], out=sys.argv[2] if len(sys.argv) > 2 else None)
