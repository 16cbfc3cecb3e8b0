#!/usr/bin/env python
"""Per-kernel table of a rocprofv3 --pmc pass (counter_collection.csv: one row per dispatch and counter):
    python tools/pmc_kernels.py <counter_collection.csv> <out.csv>
Sums every counter over the dispatches of a kernel (template arguments kept, anonymous namespace dropped), writes the raw sums as a
small CSV (committed under profiles/: the table can be recomputed from it) and prints the derived columns:
    mfma_busy / cu_busy   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES      (matrix-pipe busy cycles per busy-CU cycle, summed over SIMDs: <= 4)
    mfma_util             SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    valu / mfma           SQ_INSTS_VALU / SQ_INSTS_MFMA
    wait share            SQ_WAIT_ANY / SQ_WAVE_CYCLES"""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[n].add(r["Dispatch_Id"])
names = sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CU_CYCLES", 0.0))
counters = sorted({c for k in acc for c in acc[k]})
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches"] + counters)
    for k in names:
        w.writerow([k, len(calls[k])] + [int(acc[k].get(c, 0)) for c in counters])
print(f"{'kernel':60s} {'disp':>5s} {'mfma_busy/cu_busy':>18s} {'mfma_util':>10s} {'valu/mfma':>10s} {'wait share':>11s}")
for k in names[:40]:
    a = acc[k]
    mb, cb, gui = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), a.get("SQ_BUSY_CU_CYCLES", 0), a.get("GRBM_GUI_ACTIVE", 0)
    im, iv, wc, wa = a.get("SQ_INSTS_MFMA", 0), a.get("SQ_INSTS_VALU", 0), a.get("SQ_WAVE_CYCLES", 0), a.get("SQ_WAIT_ANY", 0)
    f = lambda x, y: f"{x / y:10.3f}" if y else f"{'-':>10s}"
    print(f"{k[:60]:60s} {len(calls[k]):5d} {f(mb, cb):>18s} {f(mb, gui / 8 * 1024)} {f(iv, im)} {f(wa, wc):>11s}")
