#!/usr/bin/env python
"""Run one GEMM shape a few times (for rocprofv3 --pmc runs):  python tools/gemm_one.py M N K [iters] [out_f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
f32 = len(sys.argv) > 5 and sys.argv[5] == "1"
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
for _ in range(iters):
    ops.gemm(a, b, out)
torch.cuda.synchronize()
print("done", M, N, K)
