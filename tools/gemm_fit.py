#!/usr/bin/env python
"""Fit the planner's cost model  time = K-tiles * c + e  (one full round of resident workgroups) for the kernel pinned by
UNITE_GEMM_KERNEL:  python tools/gemm_fit.py M N [ta tb] [bias]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unite_amd import ops
M, N = int(sys.argv[1]), int(sys.argv[2])
ta, tb = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
pts = []
for K in (512, 768, 1536, 3072, 6144):
    a = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
    b = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16 if not ta else torch.float32, device="cuda")
    bias = torch.randn(N, device="cuda") if not ta else None
    for _ in range(5):
        ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), bias=bias)
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, b, out, trans_a=bool(ta), trans_b=bool(tb), bias=bias)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 50)
    pts.append((K // 64, sorted(ts)[1]))
x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts])
c, e = np.polyfit(x, y, 1)
print(os.environ.get("UNITE_GEMM_KERNEL"), f"M={M} N={N} ta={ta} tb={tb}:", " ".join(f"{k}kt:{t:.1f}us" for k, t in pts), f"-> c={c:.3f} e={e:.2f}")
