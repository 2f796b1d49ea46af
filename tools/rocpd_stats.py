#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (like --stats CSV).

    rocpd_stats.py <db> [out.csv] [--after-marker]

--after-marker: only the launches that START after the last `k_nvfi_prof_marker` dispatch, i.e. bench.py's profiled pass (issued
on one stream; the timed region before it overlaps three streams, so its per-kernel durations include co-running kernels)."""
import sqlite3
import sys


def main(path, out=None, after_marker=False):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else cols[0]
    where = ""
    if after_marker:
        t = c.execute(f"select max(start) from kernels where {name_col} like '%k_nvfi_prof_marker%'").fetchone()[0]
        if t is None:
            raise SystemExit("no k_nvfi_prof_marker dispatch in the trace")
        where = f"where start > {int(t)}"
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels {where} group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, k, s, a, mn, mx in rows:
        lines.append(f"\"{n}\",{k},{s},{a:.1f},{mn},{mx},{100.0 * s / tot:.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(args[0], out=args[1] if len(args) > 1 else None, after_marker="--after-marker" in sys.argv)
