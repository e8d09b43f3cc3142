#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``--kernel-trace``): per-kernel launch statistics, and -- with ``--timeline N`` --
the last N dispatches with the idle gap in front of each (what the device-resident loop leaves between its kernels)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("mbar::", "").replace("void ", "")


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    stats = {}
    for name, s, e in rows:
        d = stats.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d[0] += 1
        d[1] += (e - s)
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    tot = sum(d[1] for d in stats.values())
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for k, d in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:70]:70s} {d[0]:6d} {d[1] * 1e-6:10.3f} {d[1] / d[0] * 1e-3:10.2f} {d[2] * 1e-3:10.2f} {d[3] * 1e-3:10.2f} {100 * d[1] / tot:6.2f}")
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        print("\nlast dispatches: gap before (us), duration (us), kernel")
        prev = None
        for name, s, e in rows[-n:]:
            gap = (s - prev) * 1e-3 if prev is not None else 0.0
            print(f"{gap:9.2f} {(e - s) * 1e-3:10.2f}  {short(name)[:90]}")
            prev = e


if __name__ == "__main__":
    main()
