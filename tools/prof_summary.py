#!/usr/bin/env python
"""Trim a rocprofv3 *_kernel_stats.csv into a readable per-step table:  python tools/prof_summary.py <csv> <n_steps>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':58s} {'calls/step':>10s} {'ms/step':>8s} {'avg_us':>8s} {'pct':>6s}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)[:58]
    print(f"{n:58s} {int(r['Calls'])/steps:10.1f} {float(r['TotalDurationNs'])/steps/1e6:8.3f} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
print(f"total GPU kernel time: {tot/steps/1e6:.3f} ms/step")
