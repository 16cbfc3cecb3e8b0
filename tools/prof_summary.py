#!/usr/bin/env python
"""Per-step kernel table from a rocprofv3 --kernel-trace run of bench.py:
    python tools/prof_summary.py <kernel_trace.csv> [max_rows]
Only the dispatches INSIDE the training steps are counted: the window runs from the end of the first AdamW launch to the end of the last
one (n - 1 whole steps), so one-off work -- building the flat parameter buffers (hundreds of small device copies), the first bf16 shadow
cast -- does not show up as "per step" (round 1's table divided the whole run's totals by the step count)."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
key = lambda r: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
adam = sorted(key(r)[1] for r in rows if "adamw_flat_kernel" in r["Kernel_Name"])
if len(adam) < 2:
    sys.exit("need at least two optimizer steps in the trace")
t0, t1, steps = adam[0], adam[-1], len(adam) - 1
tot, cnt = defaultdict(float), defaultdict(int)
outside = defaultdict(int)
for r in rows:
    s, e = key(r)
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    if t0 < e <= t1:
        tot[n] += e - s
        cnt[n] += 1
    else:
        outside[n] += 1
total = sum(tot.values())
print(f"# {steps} steps between the first and the last AdamW launch; kernel time {total / steps / 1e6:.3f} ms/step")
print(f"{'kernel':62s} {'calls/step':>10s} {'ms/step':>8s} {'avg_us':>8s} {'pct':>6s}")
for n in sorted(tot, key=lambda k: -tot[k])[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{n[:62]:62s} {cnt[n] / steps:10.1f} {tot[n] / steps / 1e6:8.3f} {tot[n] / cnt[n] / 1e3:8.1f} {100 * tot[n] / total:6.2f}")
setup = {k: v for k, v in outside.items() if k not in tot or v > 0}
top = sorted(setup, key=lambda k: -setup[k])[:6]
print("# outside the window (setup, first / last partial step): " + ", ".join(f"{k[:40]} x{setup[k]}" for k in top))
