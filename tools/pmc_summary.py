#!/usr/bin/env python
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd sqlite database -> CSV (name, launches, counter, total, per_launch)."""
import sqlite3
import sys
from collections import defaultdict


def main(path, out=None):
    db = sqlite3.connect(path)
    c = db.cursor()
    rows = c.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection").fetchall()
    agg = defaultdict(lambda: [0.0, set()])
    for k, d, n, v in rows:
        key = (k.split("(")[0], n)
        agg[key][0] += v
        agg[key][1].add(d)
    lines = ["Kernel,Counter,Launches,Total,PerLaunch"]
    for (k, n), (tot, ds) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"\"{k}\",{n},{len(ds)},{tot:.0f},{tot / max(len(ds), 1):.1f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
