#!/usr/bin/env python
"""Where inside a step does each HIP queue work?  For every step window (AdamW launch to AdamW launch) of an overlapped bench trace: per queue
the busy time in each tenth of the window, averaged over the steps.  python tools/trace_phases.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adamw_flat" in r["Kernel_Name"])
steps = list(zip(adam[1:-1], adam[2:]))
NB = 10
acc = collections.defaultdict(lambda: [0.0] * NB)
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    for t0, t1 in steps:
        if e <= t0 or s >= t1:
            continue
        w = (t1 - t0) / NB
        for b in range(NB):
            lo, hi = t0 + b * w, t0 + (b + 1) * w
            ov = min(e, hi) - max(s, lo)
            if ov > 0:
                acc[q][b] += ov / w
print(f"{len(steps)} steps of {sum(b - a for a, b in steps) / len(steps) / 1e6:.2f} ms; kernels in flight per queue (sum of durations / bin width), by tenth of the step")
for q, v in sorted(acc.items()):
    print(f"queue {q:>3s}: " + " ".join(f"{x / len(steps):5.2f}" for x in v))
