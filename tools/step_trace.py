#!/usr/bin/env python
"""One steady-state step of a rocprofv3 kernel trace (rocpd sqlite) as a table: start offset, duration, queue, kernel - to read how
the three streams of bench.py's timed region overlap.  usage: step_trace.py <db> [steps_from_end=6]"""
import sqlite3
import sys


def main(path, back=6, marker="k_adam"):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"name, start, end{', ' + qcol if qcol else ''}"
    rows = c.execute(f"select {sel} from kernels order by start").fetchall()
    marks = [r[2] for r in rows if r[0].startswith(marker)]
    lo, hi = marks[-back - 1], marks[-back]
    rows = [r for r in rows if lo <= r[1] < hi or (r[1] < hi and r[2] > lo and r[1] >= lo)]
    t0 = rows[0][1]
    print(f"step span {(hi - lo) / 1e3:.1f} us, {len(rows)} kernels")
    qs = {}
    for r in rows:
        q = r[3] if qcol else 0
        qs.setdefault(q, len(qs))
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{qs[q]}  {r[0][:70]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
