#!/usr/bin/env python
"""Unique (kernel, workgroup size, LDS, scratch, VGPR, AGPR, SGPR) rows of a rocprofv3 kernel trace:  python tools/trace_resources.py <kernel_trace.csv>"""
import collections, csv, re, sys
seen = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    seen[((re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+|\w+Functor\w*)", r["Kernel_Name"]) or [r["Kernel_Name"][:52]])[0][:52], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])] += 1
print(f"{'kernel':52s} {'wg':>5s} {'LDS':>7s} {'scr':>5s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'launches':>8s}")
for k, n in sorted(seen.items()):
    print(f"{k[0]:52s} {k[1]:>5s} {k[2]:>7s} {k[3]:>5s} {k[4]:>5s} {k[5]:>5s} {k[6]:>5s} {n:8d}")
