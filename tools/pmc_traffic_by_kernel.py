#!/usr/bin/env python
"""Per-kernel (name x grid size) HBM traffic of a bench run from separate FETCH_SIZE / WRITE_SIZE passes (gfx950: FETCH_SIZE x2).
python tools/pmc_traffic_by_kernel.py <fetch_csv> <write_csv>"""
import collections
import csv
import sys


def agg(path, counter, scale):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = (r["Kernel_Name"][:46], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            d[k][0] += float(r["Counter_Value"]) * 1024 * scale
            d[k][1] += 1
    return d


f = agg(sys.argv[1], "FETCH_SIZE", 2)
w = agg(sys.argv[2], "WRITE_SIZE", 1)
rows = []
for k in f:
    n = f[k][1]
    rows.append((f[k][0] + w.get(k, [0, 0])[0], k, n, f[k][0] / n / 1e6, w.get(k, [0, 1])[0] / max(w.get(k, [0, 1])[1], 1) / 1e6))
rows.sort(reverse=True)
print(f"{'kernel':48s} {'WGs':>6s} {'launches':>8s} {'fetch MB':>9s} {'write MB':>9s} {'total GB':>9s}")
for tot, k, n, fm, wm in rows[:40]:
    print(f"{k[0]:48s} {k[1]:6d} {n:8d} {fm:9.1f} {wm:9.1f} {tot / 1e9:9.2f}")
