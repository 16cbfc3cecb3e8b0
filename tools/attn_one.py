#!/usr/bin/env python
"""Run the fused attention kernels alone (for rocprofv3 --pmc):  python tools/attn_one.py B N H [iters] [bwd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unite_amd import ops
B, N, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
bwd = len(sys.argv) > 5 and sys.argv[5] == "1"
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
o = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, N, device="cuda")
do = torch.randn_like(o)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, N, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters + 2):
    if it == 2:
        e0.record()
    ops.attn_fwd(qkv, o, lse, B, N, H, 0.125)
    if bwd:
        ops.attn_bwd(qkv, o, do, lse, delta, dqkv, B, N, H, 0.125)
e1.record()
torch.cuda.synchronize()
print(f"B={B} N={N} H={H} bwd={bwd}: {e0.elapsed_time(e1) * 1e3 / iters:.1f} us per iteration")
