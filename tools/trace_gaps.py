#!/usr/bin/env python
"""GPU idle time inside the training steps of a rocprofv3 --kernel-trace run of bench.py (streams ON):
    python tools/trace_gaps.py <kernel_trace.csv>
Window: end of the first AdamW launch .. end of the last one.  Reports the union of kernel intervals (GPU busy), the idle remainder,
the serial sum (what one stream would need) and the largest idle gaps with the kernels around them."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)[:48]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r)) for r in rows)
adam = [e for s, e, n in ev if "adamw_flat_kernel" in n]
t0, t1, steps = adam[0], adam[-1], len(adam) - 1
win = [(s, e, n) for s, e, n in ev if e > t0 and s < t1]
busy, cur_s, cur_e, gaps = 0, None, None, []
last_name = ""
for s, e, n in win:
    s, e = max(s, t0), min(e, t1)
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        gaps.append((s - cur_e, last_name, n))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
serial = sum(min(e, t1) - max(s, t0) for s, e, n in win)
wall = t1 - t0
print(f"{steps} steps: wall {wall / steps / 1e6:.3f} ms/step, GPU busy (union) {busy / steps / 1e6:.3f}, idle {(wall - busy) / steps / 1e6:.3f}, "
      f"serial sum {serial / steps / 1e6:.3f} ms/step, {len(gaps) / steps:.0f} gaps/step")
from collections import defaultdict
by = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by[(a, b)][0] += g; by[(a, b)][1] += 1
print("largest idle contributors (before -> after: total us/step, count/step, avg us):")
for (a, b), (tot, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {a:48s} -> {b:48s} {tot / steps / 1e3:8.1f} {c / steps:6.1f} {tot / c / 1e3:7.2f}")
