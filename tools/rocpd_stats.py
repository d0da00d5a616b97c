#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite) kernel trace: per-kernel calls / total / average duration.
usage: python tools/rocpd_stats.py results.db [skip_first_fraction] > profiles/xxx_kernel_stats.txt"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = re.sub(r"\(.*", "", short)
        d = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    print(f"# {len(rows)} dispatches, {tot / 1e3:.3f} ms total kernel time")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:70]:70s} {v[0]:7d} {v[1] / 1e3:10.3f} {v[1] / v[0]:10.1f} {v[2]:10.1f} {v[3]:10.1f} {100 * v[1] / tot:6.2f}")


if __name__ == "__main__":
    main()
