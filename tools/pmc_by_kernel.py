#!/usr/bin/env python
"""Per kernel and launch grid: L2 fetch and write bytes per launch from the two rocprofv3 --pmc passes of tools/final_prof.sh
(FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as counted; both counters in KiB):
   python tools/pmc_by_kernel.py <fetch counter_collection.csv> <write counter_collection.csv> [rows]"""
import collections, csv, sys


def load(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return d


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k, v in f.items():
    ww = w.get(k, [0.0])
    rows.append((sum(v) * 2048 / 1e6, k, len(v), sum(v) / len(v) * 2048 / 1e6, sum(ww) / max(len(ww), 1) * 1024 / 1e6))
rows.sort(reverse=True)
tot_f = sum(r[0] for r in rows)
tot_w = sum(sum(v) for v in w.values()) * 1024 / 1e6
print(f"# all launches of the run: {tot_f / 1e3:.1f} GB fetched by the L2s (FETCH_SIZE x 2), {tot_w / 1e3:.1f} GB written")
print(f"# {'kernel':70s} {'grid':>9s} {'launches':>8s} {'fetch MB/launch':>16s} {'write MB/launch':>16s} {'share of fetch':>15s}")
for tot, (name, grid), n, fl, wl in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    name = name.split("(")[0][:70]
    print(f"  {name:70s} {grid:>9s} {n:8d} {fl:16.1f} {wl:16.1f} {100 * tot / tot_f:14.1f}%")
