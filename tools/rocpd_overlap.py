#!/usr/bin/env python
"""Overlap of the gradient all-reduce with the backward, from a rocprofv3 kernel trace (rocpd SQLite DB).
usage: python tools/rocpd_overlap.py results.db [pattern=nccl]
For every kernel whose name matches `pattern` (RCCL's device kernels): its interval and the share of it during which at
least one OTHER kernel (the backward's conv / weight-gradient / BatchNorm kernels on the compute stream) was executing."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else "nccl|rccl", re.I)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    coll = [(s, e, n) for n, s, e in rows if pat.search(n)]
    other = [(s, e) for n, s, e in rows if not pat.search(n)]
    print(f"# {len(rows)} dispatches, {len(coll)} collective kernels matching /{pat.pattern}/")
    if not coll:
        names = sorted({re.sub(r'\(.*', '', n)[:60] for n, _, _ in rows})
        print("# no collective kernel in the trace; kernel names seen:", ", ".join(names[:12]), "...")
        return
    tot = ov = 0.0
    j = 0
    for s, e, n in coll:
        covered, cur = 0.0, s
        while j < len(other) and other[j][1] <= s:
            j += 1
        k = j
        while k < len(other) and other[k][0] < e:
            a, b = max(other[k][0], cur), min(other[k][1], e)
            if b > a:
                covered += b - a
                cur = b
            k += 1
        tot += e - s
        ov += covered
    first, last = coll[0][0], coll[-1][1]
    print(f"collective kernels: {len(coll)} launches, {tot / 1e3:.1f} us total, {ov / 1e3:.1f} us ({100 * ov / max(tot, 1):.1f} %) "
          f"concurrent with compute kernels; first starts {(first - rows[0][1]) / 1e6:.2f} ms into the trace, last ends at "
          f"{(last - rows[0][1]) / 1e6:.2f} ms")
    agg = {}
    for s, e, n in coll:
        k = re.sub(r"\(.*", "", n)[:70]
        d = agg.setdefault(k, [0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3
    for k, (n, us) in agg.items():
        print(f"  {k:70s} {n:5d} launches {us / n:9.1f} us avg")


if __name__ == "__main__":
    main()
