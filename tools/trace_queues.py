#!/usr/bin/env python
"""Per HIP queue of a rocprofv3 kernel trace of bench.py: kernels per step, busy time (union of its kernel intervals), sum of kernel durations
and the share of the step window the queue spends waiting.  python tools/trace_queues.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adamw_flat" in r["Kernel_Name"])
t0, t1, steps = adam[0], adam[-1], len(adam) - 1
q = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0 and e <= t1:
        q[r["Queue_Id"]].append((s, e, r["Kernel_Name"]))
print(f"{steps} steps, window {(t1 - t0) / steps / 1e6:.3f} ms/step")
for k, iv in sorted(q.items(), key=lambda kv: -len(kv[1])):
    iv.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _ in iv)
    names = collections.Counter(n.split("(")[0][-40:] for _, _, n in iv).most_common(3)
    print(f"queue {k:>3s}: {len(iv) / steps:6.1f} kernels/step  busy {busy / steps / 1e6:7.3f} ms/step  kernel sum {tot / steps / 1e6:7.3f}  "
          f"mostly {', '.join(n for n, _ in names)}")
