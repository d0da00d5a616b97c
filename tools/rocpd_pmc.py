#!/usr/bin/env python
"""Per-kernel PMC summary from rocprofv3 rocpd DBs (one DB per --pmc pass).
usage: python tools/rocpd_pmc.py fetch_results.db write_results.db
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md section HBM): FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced reads -> the read side is doubled ("fetch_x2" column); WRITE_SIZE is uncalibrated."""
import re
import sqlite3
import sys


def load(path):
    c = sqlite3.connect(path).cursor()
    out = {}
    for name, cname, val in c.execute("select name, counter_name, counter_value from pmc_events"):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = re.sub(r"\(.*", "", short)
        d = out.setdefault((short, cname), [0, 0.0])
        d[0] += 1
        d[1] += val
    return out


def dump(paths, pattern):
    """--raw PATTERN db...: every counter summed over dispatches for kernels matching PATTERN (per-call averages)"""
    for p in paths:
        for (k, cn), (n, v) in sorted(load(p).items()):
            if pattern in k:
                print(f"{k[:50]:50s} {cn:32s} calls {n:4d}  per-call {v / max(n, 1):16.1f}")


def to_json(out_path, paths):
    """--json OUT db...: profiles/pmc_traffic.json (HBM bytes per launch, FETCH_SIZE doubled = gfx950 correction)"""
    import json
    agg = {}
    for p in paths:
        for (k, cn), (n, v) in load(p).items():
            agg.setdefault(k, {})[cn] = (n, v)
    kernels = {}
    for k, d in agg.items():
        f = d.get("FETCH_SIZE", (1, 0.0)); w = d.get("WRITE_SIZE", (1, 0.0))
        fb = 2.0 * f[1] / max(f[0], 1) * 1024; wb = w[1] / max(w[0], 1) * 1024
        kernels[k.replace(", ", ",")] = {"launches": max(f[0], w[0]), "fetch_bytes_per_launch_x2_corrected": fb,
                                          "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 1 (bs16 train step); "
            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); calibration: bn_gelu_apply "
            "2*FETCH == WRITE, the canvas zero-fill WRITE == 16*512*512*64*4 B")
    with open(out_path, "w") as fh:
        json.dump({"_note": note, "kernels": kernels}, fh, indent=1)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--raw":
        return dump(sys.argv[3:], sys.argv[2])
    if len(sys.argv) > 3 and sys.argv[1] == "--json":
        return to_json(sys.argv[2], sys.argv[3:])
    agg = {}
    for p in sys.argv[1:]:
        for (k, cn), (n, v) in load(p).items():
            agg.setdefault(k, {})[cn] = (n, v)
    print(f"{'kernel':44s} {'calls':>6s} {'FETCH_MB/call':>14s} {'fetch_x2_MB':>12s} {'WRITE_MB/call':>14s}")
    rows = []
    for k, d in agg.items():
        n = max(v[0] for v in d.values())
        f = d.get("FETCH_SIZE", (1, 0.0)); w = d.get("WRITE_SIZE", (1, 0.0))
        rows.append((k, n, f[1] / max(f[0], 1) * 1024 / 1e6, w[1] / max(w[0], 1) * 1024 / 1e6))
    for k, n, f, w in sorted(rows, key=lambda r: -(r[2] + r[3]) * r[1])[:40]:
        print(f"{k[:44]:44s} {n:6d} {f:14.2f} {2 * f:12.2f} {w:14.2f}")


if __name__ == "__main__":
    main()
